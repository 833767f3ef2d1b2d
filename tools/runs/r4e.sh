#!/bin/bash
# round 4, call 5: two accumulator chains per sub-tile in the S waves, deeper fragment prefetch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
( timeout 120 python tools/hstu_fwd_ab.py --shapes c3,d4096,d8x4096,jag1,ragged
  for v in k2v2 k3v2 k2v3 k4v4 k3v3sp4; do timeout 120 env MI355_LIB=$L/librecsys_amd_$v.so python tools/hstu_fwd_ab.py --shapes c3,d4096,d8x4096,jag1; done 
  timeout 120 python tools/hstu_fwd_ab.py --shapes c3,d4096,d8x4096,jag1 ) > $O/ab.txt 2>&1
grep -v amdgpu.ids $O/ab.txt
timeout 600 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
( for v in tim tpr2 tpr8; do echo "== $v"; MI355_LIB=$L/librecsys_amd_$v.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 8 --seqlen 4096; done ) > $O/stamps.txt 2>&1
grep -v amdgpu.ids $O/stamps.txt
