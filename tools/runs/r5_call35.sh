#!/bin/bash
# round 5, call 35b: steady-state partition kernel with / without the block-wide row initialisation code in the eviction path
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cd /tmp && export TMPDIR=/tmp
for v in default nofresh default nofresh; do
  if [ $v != default ]; then export MI355_LIB=$R/recsys-examples_amd/lib/librecsys_amd_$v.so; else unset MI355_LIB; fi
  rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/kt.log 2>&1
  DB=$(find /tmp/kt -name '*.db' | head -1)
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' /tmp/kt.log | head -1)"; python $R/tools/rocpd_stats.py $DB | grep "part3" | cut -c1-60,72-140
done
