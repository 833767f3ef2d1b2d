#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
S=c3,d32x256,d32x768,d1024,d16x2048,d8x4096,d4096,jag1,jag2,jag3,jag4,ragged
( for q in 0 1 0 1; do echo "== Q2=$q"; MI355_HSTU_Q2=$q timeout 300 python tools/hstu_fwd_ab.py --shapes $S 2>&1 | grep -v amdgpu.ids; done ) > $O/ab.txt 2>&1; cat $O/ab.txt
