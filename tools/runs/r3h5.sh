#!/bin/bash
mkdir -p gpurun_out/r3h5
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  MI355_HSTU_DS=$1 MI355_HSTU_XP=$2 timeout 300 python tools/hstu_shapes.py --seeds 1 > gpurun_out/r3h5/ds$1_xp$2.txt 2>&1
  echo "DS=$1 XP=$2"; grep -v amdgpu gpurun_out/r3h5/ds$1_xp$2.txt | cut -c1-20,58-140
done
