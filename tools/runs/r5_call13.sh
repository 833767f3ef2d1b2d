#!/bin/bash
# round 5, call 13: streaming (nontemporal) stores for the outputs of the probe kernel (dbg 16) / the partition kernel (dbg 32):
# does the end-of-kernel L2 write-back shrink?  kernel trace per setting
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c13; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
cd /tmp && export TMPDIR=/tmp
for c in 0 16 32 48 0; do
MI355_FUSED_DBG=$c timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt$c -o t -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt$c.log 2>&1
DB=$(find /tmp/kt$c -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kt_dbg$c.txt; echo "== dbg $c"; head -5 $O/kt_dbg$c.txt | cut -c1-60,72-140
grep -o '"ms_per_step": [0-9.]*' /tmp/kt$c.log | head -1
rm -rf /tmp/kt$c
done
