#!/bin/bash
mkdir -p gpurun_out/r3h8
timeout 1200 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > gpurun_out/r3h8/tests.txt 2>&1; tail -2 gpurun_out/r3h8/tests.txt
for i in 1 2; do
  for v in 1 0; do MI355_HSTU_VQ=$v timeout 300 python tools/hstu_shapes.py --seeds 2 > gpurun_out/r3h8/vq${v}_$i.txt 2>&1; done
done
for f in vq1_1 vq0_1 vq1_2 vq0_2; do echo $f; grep -v "amdgpu\|MI355" gpurun_out/r3h8/$f.txt | cut -c1-20,100-140; done
