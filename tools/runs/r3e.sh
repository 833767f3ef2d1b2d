#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 30; do
  rm -rf /tmp/kt
  MI355_POOL_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt.log 2>&1
  DB=$(find /tmp/kt -name '*.db' | head -1)
  python $R/tools/rocpd_stats.py $DB | head -7 | cut -c1-150 > $O/stats_v$v.txt
  python $R/tools/rocpd_timeline.py $DB 12 | cut -c1-120 > $O/timeline_v$v.txt
  cat $O/stats_v$v.txt
done
cd $R
MI355_LIB=$R/recsys-examples_amd/lib/librecsys_amd.so python tools/bench_gather_c2.py 20 > $O/gather_standalone.txt 2>&1
MI355_POOL_VARIANT=30 python tools/bench_gather_c2.py 20 >> $O/gather_standalone.txt 2>&1
cat $O/gather_standalone.txt
