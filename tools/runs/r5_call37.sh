#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c37; mkdir -p $O; cd $R
for i in $(seq 1 14); do
timeout 300 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -x -k "around_its_size_limits or big_batches or evict or full" > $O/pytest_$i.txt 2>&1; grep "passed\|failed" $O/pytest_$i.txt | tr '\n' ' '
if grep -q "failed" $O/pytest_$i.txt; then grep -B40 "Error\|assert " $O/pytest_$i.txt | head -150; break; fi
done
