#!/bin/bash
mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -x -q -m gpu -k "around_its_size_limits" > gpurun_out/r3t/tests.txt 2>&1
tail -30 gpurun_out/r3t/tests.txt
