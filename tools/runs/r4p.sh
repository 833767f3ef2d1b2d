#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "" _xp1 _xp2 _xp3 _xp4; do
  for cfg in "32 512" "8 4096"; do
    set -- $cfg
    rm -rf /tmp/prof
    MI355_LIB=$R/recsys-examples_amd/lib/librecsys_amd$v.so rocprofv3 --kernel-trace -d /tmp/prof -o t -- python $R/tools/bench_hstu.py --batch $1 --seqlen $2 --reps 6 > /dev/null 2>&1
    db=$(find /tmp/prof -name '*.db' | head -1)
    echo "== lib$v B $1 L $2: $(python $R/tools/rocpd_stats.py $db | grep -i 'v_p8' | awk '{print $(NF-3)}') us (v_p8 avg)"
  done
done > $O/probe.txt 2>&1
cat $O/probe.txt
