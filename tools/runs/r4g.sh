#!/bin/bash
# round 4, call 7: the P / dS exchange of the attention backward under a byte cap (jagged, chunked)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu -k "exchange or scratch or golden or window or fp16" > $O/tests.txt 2>&1; tail -8 $O/tests.txt
timeout 300 python tools/hstu_shapes.py --seeds 2 > $O/shapes.txt 2>&1; grep -v amdgpu.ids $O/shapes.txt
MI355_HSTU_DS_MAX_BYTES=17179869184 timeout 300 python tools/hstu_shapes.py --seeds 2 > $O/shapes_dense.txt 2>&1; grep -v amdgpu.ids $O/shapes_dense.txt
