#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4ae; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
S=d16x2048,d8x4096,d4096,jag1
( for v in "" _q2pk "" _q2pk; do echo "== lib$v"; MI355_LIB=$L/librecsys_amd$v.so timeout 300 python tools/hstu_fwd_ab.py --shapes $S 2>&1 | grep -v amdgpu.ids; done ) > $O/ab.txt 2>&1; cat $O/ab.txt
