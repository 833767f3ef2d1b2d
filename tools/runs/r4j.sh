#!/bin/bash
# round 4: per-kernel times of the attention kernels (rocprofv3 kernel trace) at C3 and 8 x 4096, x8 on / off
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "c3 32 512 1" "l4096 8 4096 1"; do
  set -- $cfg
  rm -rf /tmp/prof_$1
  MI355_HSTU_X8=$4 rocprofv3 --kernel-trace -d /tmp/prof_$1 -o t -- python $R/tools/bench_hstu.py --batch $2 --seqlen $3 --reps 10 > $O/$1.log 2>&1
  db=$(find /tmp/prof_$1 -name '*.db' | head -1)
  echo "== $1 (X8=$4)"; python $R/tools/rocpd_stats.py $db | grep -i "hstu\|total" | cut -c1-150
done > $O/kernels.txt 2>&1
cat $O/kernels.txt
