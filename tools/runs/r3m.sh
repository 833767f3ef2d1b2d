#!/bin/bash
mkdir -p gpurun_out/r3m
timeout 600 python tools/bench_model_shapes.py > gpurun_out/r3m/out.txt 2>&1; grep -v amdgpu gpurun_out/r3m/out.txt | tail -12
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/r3m/prof -o t -- python $R/tools/bench_model_shapes.py --steps 20 > /dev/null 2>&1
db=$(find $R/gpurun_out/r3m/prof -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $db 60 > $R/gpurun_out/r3m/timeline.txt
python $R/tools/rocpd_stats.py $db > $R/gpurun_out/r3m/stats.txt
rm -rf $R/gpurun_out/r3m/prof
head -25 $R/gpurun_out/r3m/stats.txt | cut -c1-150
