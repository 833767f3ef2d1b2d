#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c34; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1500 python -m pytest tests/test_hstu_gpu.py -q -m gpu > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B12 "Error\|assert " $O/pytest_a.txt | head -40
