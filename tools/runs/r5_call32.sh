#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c32; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 600 python -m pytest tests/test_sharded_gpu.py -q -m gpu -x -k "empty" > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B14 "Error\|assert " $O/pytest_a.txt | head -70
