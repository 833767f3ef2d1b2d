#!/bin/bash
# HSTU block rotation A/B
mkdir -p gpurun_out/r3h1
for r in 0 1 4 5; do
  MI355_HSTU_ROT=$r timeout 300 python tools/hstu_shapes.py > gpurun_out/r3h1/rot$r.txt 2>&1
done
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > gpurun_out/r3h1/tests.txt 2>&1
tail -3 gpurun_out/r3h1/tests.txt
paste -d'\n' gpurun_out/r3h1/rot0.txt gpurun_out/r3h1/rot1.txt | grep -v amdgpu.ids
