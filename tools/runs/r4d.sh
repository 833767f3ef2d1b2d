#!/bin/bash
# round 4, call 4: timing probes of hstu_fwd_pc_kernel (results wrong on purpose): what bounds the 3.5 K cycles per tile
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
( timeout 120 python tools/hstu_fwd_ab.py --shapes c3,d8x4096
  for v in pr1 pr2 pr4 pr8 pr16 pr6 pr10 pr24 pr30; do timeout 120 env MI355_LIB=$L/librecsys_amd_$v.so python tools/hstu_fwd_ab.py --shapes c3,d8x4096; done ) > $O/ab.txt 2>&1
grep -v amdgpu.ids $O/ab.txt
( for v in tpr1 tpr2 tpr4 tpr8; do echo "== $v"; MI355_LIB=$L/librecsys_amd_$v.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 8 --seqlen 4096; done ) > $O/stamps.txt 2>&1
grep -v amdgpu.ids $O/stamps.txt
