#!/bin/bash
# round 5, call 26: eviction scan restructured -- eviction tests, then the step while the table fills (steps 300)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c26; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py tests/test_twin_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B12 "Error\|assert " $O/pytest_a.txt | head -50
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kt300.txt; head -6 $O/kt300.txt | cut -c1-60,72-140
python $R/tools/rocpd_timeline.py $DB 8 | cut -c1-110
grep -o '"ms_per_step": [0-9.]*' /tmp/kt.log | head -2
