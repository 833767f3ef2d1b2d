#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3i; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd $R
MI355_LIB=$L/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py 2>&1 | grep -v "^==.*probe" | sed -n '/part2 blocks/,$p' > $O/stamps_kit7.txt
cat $O/stamps_kit7.txt
cd /tmp && export TMPDIR=/tmp
for v in p2nopart p2nogather; do
  rm -rf /tmp/kt
  MI355_LIB=$L/librecsys_amd_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt.log 2>&1
  DB=$(find /tmp/kt -name '*.db' | head -1)
  echo "== $v"; python $R/tools/rocpd_stats.py $DB | head -5 | cut -c1-150
done
