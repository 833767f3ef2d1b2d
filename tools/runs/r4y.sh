#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
( for c in 1073741824 2147483648 4294967296 8589934592 17179869184 1073741824 4294967296; do echo "== MI355_HSTU_DS_MAX_BYTES=$c"; MI355_HSTU_DS_MAX_BYTES=$c timeout 300 python tools/hstu_shapes.py --seeds 1 2>&1 | grep "x 4096\|seed 1\|uniform"; done ) > $O/cap.txt 2>&1; cat $O/cap.txt
