#!/bin/bash
mkdir -p gpurun_out/r3t2
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -x -q -m gpu -k "multi_table_batches" > gpurun_out/r3t2/tests.txt 2>&1
tail -25 gpurun_out/r3t2/tests.txt
