#!/bin/bash
# round 5, call 18: kernel timeline of the forced W = 1 sharded step (in-library exchange)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c18; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/sh -o t -- python $R/tools/runs/sharded_w1_loop.py > $O/loop.log 2>&1; grep "ms/step" $O/loop.log
db=$(find /tmp/sh -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $db 60 > $O/sharded_w1_timeline.txt
python $R/tools/rocpd_stats.py $db > $O/sharded_w1_stats.txt
cat $O/sharded_w1_timeline.txt | tail -45; head -20 $O/sharded_w1_stats.txt
cd $R
for v in 1 0; do MI355_EXCHANGE_PRIO=$v python tools/runs/sharded_w1_loop.py 2>&1 | grep "ms/step"; done
