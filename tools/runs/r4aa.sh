#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4aa; mkdir -p $O; cd $R
( for x in 0 1 0 1; do echo "== MI355_HSTU_XE=$x"; MI355_HSTU_XE=$x timeout 300 python tools/hstu_shapes.py --seeds 1 2>&1 | grep "x 4096\|seed 1\|uniform"; done ) > $O/xe.txt 2>&1; cat $O/xe.txt
MI355_HSTU_XE=2 MI355_HSTU_CHILD=1 timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu -k "256 and not mask_alone and not rab and not kernel_variants" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu -k "scratch or chunks" 2>&1 | tail -3
