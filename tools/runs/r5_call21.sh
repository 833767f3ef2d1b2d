#!/bin/bash
# round 5, call 21: mask functions (func) inside the attention kernels -- func tests, whole attention suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c21; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_hstu_gpu.py -q -m gpu -x -k "mask or func or arbitrary" > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B12 "Error\|assert " $O/pytest_a.txt | head -70
timeout 1200 python -m pytest tests/test_hstu_gpu.py tests/test_plugin_surface_gpu.py -q -m gpu > $O/pytest_b.txt 2>&1; grep "passed\|failed" $O/pytest_b.txt
