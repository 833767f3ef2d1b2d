#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
MI355_LIB=$R/recsys-examples_amd/lib/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py > $O/stamps_c.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB | head -8 | cut -c1-150 > $O/stats_c.txt
python $R/tools/rocpd_timeline.py $DB 12 | cut -c1-140 > $O/timeline_c.txt
cat $O/stats_c.txt; cat $O/timeline_c.txt | head -14
cat $O/stamps_c.txt
