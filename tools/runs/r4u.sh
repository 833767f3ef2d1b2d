#!/bin/bash
# q2 forward: finer S-wave stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
S=c3,d4096,d8x4096,jag1,ragged
( for q in 0 1 0 1; do echo "== Q2=$q"; MI355_HSTU_Q2=$q timeout 300 python tools/hstu_fwd_ab.py --shapes $S 2>&1 | grep -v amdgpu.ids; done ) > $O/ab.txt 2>&1; cat $O/ab.txt
( for v in tq2; do echo "== $v"; MI355_HSTU_PAIR=0 MI355_LIB=$L/librecsys_amd_$v.so timeout 200 python tools/hstu_phase_cycles.py --q2 --batch 8 --seqlen 4096 2>&1 | grep -v amdgpu.ids; done ) > $O/stamps.txt 2>&1; cat $O/stamps.txt
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
