#!/bin/bash
# one-GEMM backward passes with one part removed at a time (results wrong on purpose): per-kernel times from rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4z; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd /tmp && export TMPDIR=/tmp
for v in "" _xp2 _xp6 _xp7; do
  rm -rf /tmp/prof_x
  MI355_LIB=$L/librecsys_amd$v.so rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o t -- python $R/tools/bench_hstu.py --batch 8 --seqlen 4096 --reps 6 > /dev/null 2>&1
  db=$(find /tmp/prof_x -name '*.db' | head -1)
  echo "== lib$v"; python $R/tools/rocpd_stats.py $db | grep -i "hstu" | cut -c1-150
done > $O/probe.txt 2>&1
cat $O/probe.txt
