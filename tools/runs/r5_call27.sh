#!/bin/bash
# round 5, call 27: eviction chain shortened (keys of deferred records in LDS, lane minimum first) -- tests, steps 300 trace, steps 50 line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c27; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py tests/test_twin_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B12 "Error\|assert " $O/pytest_a.txt | head -50
for st in 50 300; do
timeout 300 python bench.py --steps $st --warmup 10 --no-cpu-baseline --no-hstu --no-extra --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $st ms', round(d['ms_per_step'],5), 'sus', round(d['sustained']['ms_per_step'],5), {k: v.get('ms_per_step') for k, v in d['model_shapes'].items() if isinstance(v, dict)})"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kt300.txt; head -6 $O/kt300.txt | cut -c1-60,72-140
python $R/tools/rocpd_timeline.py $DB 8 | cut -c1-110
