#!/bin/bash
mkdir -p gpurun_out/r3h10
timeout 1200 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu -k "block_map or c3_full or c4_jagged" > gpurun_out/r3h10/tests.txt 2>&1; tail -4 gpurun_out/r3h10/tests.txt
