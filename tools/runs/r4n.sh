#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu -k "exchange or golden or random_jagged or scratch" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
( for x in 4 8; do echo "== X8=$x"; MI355_HSTU_X8=$x timeout 300 python tools/hstu_shapes.py --seeds 1 2>&1 | grep -v amdgpu; done ) > $O/shapes.txt 2>&1; cat $O/shapes.txt
