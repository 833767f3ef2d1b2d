#!/bin/bash
mkdir -p gpurun_out/r3h6
L=$PWD/recsys-examples_amd/lib/librecsys_amd_hstutime.so
for cfg in "32 512" "32 4096" "8 4096"; do
  set -- $cfg
  echo "== batch $1 seqlen $2 (dma)"; MI355_LIB=$L timeout 200 python tools/hstu_phase_cycles.py --dma --batch $1 --seqlen $2 2>&1 | grep -v amdgpu
done > gpurun_out/r3h6/fwd_dma.txt 2>&1
echo "== bwd C3" >> gpurun_out/r3h6/fwd_dma.txt; MI355_LIB=$L timeout 200 python tools/hstu_phase_cycles.py --bwd 2>&1 | grep -v amdgpu >> gpurun_out/r3h6/fwd_dma.txt
cat gpurun_out/r3h6/fwd_dma.txt
