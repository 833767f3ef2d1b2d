#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4r; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
( for v in "" _ks4 _ks8 _ks12 _ks16; do echo "== lib$v"; MI355_LIB=$L/librecsys_amd$v.so timeout 300 python tools/hstu_shapes.py --seeds 1 2>&1 | grep "C3\|8 x 4096\|seed 1"; done ) > $O/shapes.txt 2>&1; cat $O/shapes.txt
