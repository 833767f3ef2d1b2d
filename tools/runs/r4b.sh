#!/bin/bash
# round 4, call 2: S wave software-pipelined (SiLU of sub-tile 0 under the MFMAs of sub-tile 1), mask modes, variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
( timeout 120 python tools/hstu_fwd_ab.py --shapes c3,d4096,d8x4096,jag1,ragged
  for v in pk sgb s2 pks2 p2; do timeout 120 env MI355_LIB=$L/librecsys_amd_$v.so python tools/hstu_fwd_ab.py --shapes c3,d4096,d8x4096,jag1; done ) > $O/ab.txt 2>&1
grep -v amdgpu.ids $O/ab.txt
timeout 600 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
( MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 8 --seqlen 4096
  MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 32 --seqlen 512 ) > $O/stamps.txt 2>&1
grep -v amdgpu.ids $O/stamps.txt
