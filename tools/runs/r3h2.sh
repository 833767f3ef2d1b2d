#!/bin/bash
mkdir -p gpurun_out/r3h2
for i in 1 2 3; do
  for r in 0 4 auto; do
    if [ $r = auto ]; then unset MI355_HSTU_ROT; else export MI355_HSTU_ROT=$r; fi
    timeout 300 python tools/hstu_shapes.py --seeds 1 > gpurun_out/r3h2/rot${r}_$i.txt 2>&1
  done
done
unset MI355_HSTU_ROT
timeout 300 python tools/hstu_shapes.py --heads 8 --seeds 2 > gpurun_out/r3h2/h8_auto.txt 2>&1
MI355_HSTU_ROT=0 timeout 300 python tools/hstu_shapes.py --heads 8 --seeds 2 > gpurun_out/r3h2/h8_rot0.txt 2>&1
timeout 300 python tools/hstu_shapes.py --heads 2 --seeds 2 > gpurun_out/r3h2/h2_auto.txt 2>&1
MI355_HSTU_ROT=0 timeout 300 python tools/hstu_shapes.py --heads 2 --seeds 2 > gpurun_out/r3h2/h2_rot0.txt 2>&1
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > gpurun_out/r3h2/tests.txt 2>&1
tail -2 gpurun_out/r3h2/tests.txt
grep -h "C3 dense\|dense 32\|seed 1" gpurun_out/r3h2/rot*.txt | sort | cut -c1-130
