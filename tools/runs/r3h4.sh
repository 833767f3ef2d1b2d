#!/bin/bash
mkdir -p gpurun_out/r3h4
for v in 0 1; do
  MI355_HSTU_XDMA=$v timeout 300 python tools/hstu_shapes.py --seeds 1 > gpurun_out/r3h4/xdma$v.txt 2>&1
done
MI355_HSTU_XDMA=1 timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > gpurun_out/r3h4/tests.txt 2>&1
tail -2 gpurun_out/r3h4/tests.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MI355_HSTU_XDMA=1 rocprofv3 --kernel-trace -d $R/gpurun_out/r3h4/prof -o t -- python $R/tools/hstu_shapes.py --seeds 1 > /dev/null 2>&1
db=$(find $R/gpurun_out/r3h4/prof -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $db > $R/gpurun_out/r3h4/stats.txt
rm -rf $R/gpurun_out/r3h4/prof
cd $R
for v in 0 1; do echo xdma$v; grep -v amdgpu gpurun_out/r3h4/xdma$v.txt | cut -c1-20,58-140; done
grep "hstu_bwd" gpurun_out/r3h4/stats.txt | cut -c1-150
