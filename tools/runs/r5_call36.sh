#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c36; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
for st in 100 300; do
  rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps $st --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/kt.log 2>&1
  DB=$(find /tmp/kt -name '*.db' | head -1)
  echo "== steps $st $(grep -o '"ms_per_step": [0-9.]*' /tmp/kt.log | head -1)"; python $R/tools/rocpd_stats.py $DB | grep "part3\|probe_c" | cut -c1-60,72-140
done
cd $R; timeout 600 python -m pytest tests/test_fused_fwd_gpu.py tests/test_twin_gpu.py -q -m gpu -x 2>&1 | grep "passed\|failed"
