#!/bin/bash
# round 5, call 19: host profile of the forced W = 1 sharded step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c19; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 300 python tools/runs/prof_sharded_host.py > $O/prof_host.txt 2>&1; grep -v "^$" $O/prof_host.txt | head -40 | cut -c1-60,120-190
