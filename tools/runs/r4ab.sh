#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4ab; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
( for v in _nt0 "" _nt0 ""; do echo "== lib$v"; MI355_LIB=$L/librecsys_amd$v.so timeout 300 python tools/hstu_shapes.py --seeds 1 2>&1 | grep "C3\|x 4096\|seed 1\|uniform"; done ) > $O/nt.txt 2>&1; cat $O/nt.txt
