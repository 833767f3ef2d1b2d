#!/bin/bash
mkdir -p gpurun_out/r3x16
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r3x16/prof -o t -- python $R/bench.py --batch 1048576 --steps 6 --warmup 2 --no-hstu --no-cpu-baseline --no-extra --no-kernel-timing > $R/gpurun_out/r3x16/bench.json 2> $R/gpurun_out/r3x16/err.txt
db=$(find $R/gpurun_out/r3x16/prof -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $db > $R/gpurun_out/r3x16/stats.txt
python $R/tools/rocpd_timeline.py $db 40 > $R/gpurun_out/r3x16/timeline.txt
rm -rf $R/gpurun_out/r3x16/prof
head -12 $R/gpurun_out/r3x16/stats.txt | cut -c1-150
