#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -x -q -m gpu > gpurun_out/r3u/tests_default.txt 2>&1; tail -2 gpurun_out/r3u/tests_default.txt
MI355_FUSED_PART=1 timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -k "not around_its_size_limits" > gpurun_out/r3u/tests_part1.txt 2>&1; tail -5 gpurun_out/r3u/tests_part1.txt
