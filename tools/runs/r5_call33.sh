#!/bin/bash
# round 5, call 33: eviction without the re-probe / without the drain in the standalone partition kernel -- tests, filling regime
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c33; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1200 python -m pytest tests/test_fused_fwd_gpu.py tests/test_twin_gpu.py tests/test_module_gpu.py tests/test_demb_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B12 "Error\|assert " $O/pytest_a.txt | head -50
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kt300.txt; head -6 $O/kt300.txt | cut -c1-60,72-140
python $R/tools/rocpd_timeline.py $DB 8 | cut -c1-110
cd $R
timeout 600 python bench.py --no-cpu-baseline --no-hstu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', round(d['ms_per_step'],5), 'filling', d['c2_table_filling'])"
