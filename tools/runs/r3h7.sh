#!/bin/bash
mkdir -p gpurun_out/r3h7
L=$PWD/recsys-examples_amd/lib
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > gpurun_out/r3h7/tests.txt 2>&1
tail -2 gpurun_out/r3h7/tests.txt
for i in 1 2; do
  timeout 300 python tools/hstu_shapes.py --seeds 1 > gpurun_out/r3h7/spread_$i.txt 2>&1
  MI355_LIB=$L/librecsys_amd_nospread.so timeout 300 python tools/hstu_shapes.py --seeds 1 > gpurun_out/r3h7/nospread_$i.txt 2>&1
done
for f in spread_1 nospread_1 spread_2 nospread_2; do echo $f; grep -v amdgpu gpurun_out/r3h7/$f.txt | cut -c1-20,58-100; done
MI355_LIB=$L/librecsys_amd_hstutime.so timeout 200 python tools/hstu_phase_cycles.py --dma --batch 8 --seqlen 4096 2>&1 | grep -v amdgpu > gpurun_out/r3h7/stamps.txt; cat gpurun_out/r3h7/stamps.txt
