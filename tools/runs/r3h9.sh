#!/bin/bash
mkdir -p gpurun_out/r3h9
timeout 1200 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > gpurun_out/r3h9/tests.txt 2>&1; tail -2 gpurun_out/r3h9/tests.txt
for i in 1 2; do
  for v in 1 0; do MI355_HSTU_CM=$v timeout 300 python tools/hstu_shapes.py --seeds 1 > gpurun_out/r3h9/cm${v}_$i.txt 2>&1; done
done
for f in cm1_1 cm0_1 cm1_2 cm0_2; do echo $f; grep -v "amdgpu\|MI355" gpurun_out/r3h9/$f.txt | cut -c1-20,58-140; done
