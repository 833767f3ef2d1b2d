#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4x; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
S=d16x2048,d8x4096,d4096,jag1
( for v in "" _q2pk "" _q2pk; do echo "== lib$v"; MI355_LIB=$L/librecsys_amd$v.so timeout 300 python tools/hstu_fwd_ab.py --shapes $S 2>&1 | grep -v amdgpu.ids; done ) > $O/ab.txt 2>&1; cat $O/ab.txt
( for v in tim tq2pk; do echo "== $v"; MI355_HSTU_PAIR=0 MI355_LIB=$L/librecsys_amd_$v.so timeout 200 python tools/hstu_phase_cycles.py --q2 --batch 8 --seqlen 4096 2>&1 | grep -v amdgpu.ids; done ) > $O/stamps.txt 2>&1; cat $O/stamps.txt
