#!/bin/bash
mkdir -p gpurun_out/r3f16
timeout 1200 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > gpurun_out/r3f16/tests.txt 2>&1
tail -15 gpurun_out/r3f16/tests.txt
