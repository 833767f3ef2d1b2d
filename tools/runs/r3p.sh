#!/bin/bash
mkdir -p gpurun_out/r3p
timeout 900 python -m pytest tests/test_twin_gpu.py tests/test_module_gpu.py -x -q -m gpu -k "prefetch or admission or tiers" > gpurun_out/r3p/tests.txt 2>&1
tail -25 gpurun_out/r3p/tests.txt
