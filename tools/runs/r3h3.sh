#!/bin/bash
mkdir -p gpurun_out/r3h3
for i in 1 2; do
  MI355_HSTU_DMA=0 timeout 300 python tools/hstu_shapes.py --seeds 2 > gpurun_out/r3h3/dma0_$i.txt 2>&1
  MI355_HSTU_DMA=1 timeout 300 python tools/hstu_shapes.py --seeds 2 > gpurun_out/r3h3/dma1_$i.txt 2>&1
done
MI355_HSTU_DMA=1 timeout 900 python -m pytest tests/test_hstu_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > gpurun_out/r3h3/tests_dma1.txt 2>&1
tail -2 gpurun_out/r3h3/tests_dma1.txt
for f in dma0_1 dma1_1 dma0_2 dma1_2; do echo $f; grep -v amdgpu gpurun_out/r3h3/$f.txt | cut -c1-20,58-100; done
