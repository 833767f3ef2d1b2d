#!/bin/bash
# round 4: multi-table batches on path (c) -- the fused-forward tests, then the model shapes with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4mt; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -x > $O/pytest_fused.txt 2>&1; tail -15 $O/pytest_fused.txt
timeout 300 python tools/bench_model_shapes.py --steps 50 > $O/shapes_mt.txt 2>&1; cat $O/shapes_mt.txt | cut -c1-200
MI355_FUSED_MT=0 timeout 300 python tools/bench_model_shapes.py --steps 50 > $O/shapes_mt0.txt 2>&1; cat $O/shapes_mt0.txt | cut -c1-200
