#!/bin/bash
# forward with 64 query rows per wave (hstu_fwd_q2_kernel): checksums + times against the 32-row kernels, stamps, the forward tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4t; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
S=c3,d4096,d8x4096,jag1,jag2,ragged,d1024
( for q in 0 1 0 1; do echo "== MI355_HSTU_Q2=$q"; MI355_HSTU_Q2=$q timeout 300 python tools/hstu_fwd_ab.py --shapes $S 2>&1 | grep -v amdgpu.ids; done ) > $O/ab.txt 2>&1; cat $O/ab.txt
( MI355_LIB=$L/librecsys_amd_tq2.so timeout 200 python tools/hstu_phase_cycles.py --q2 --batch 8 --seqlen 4096 ) > $O/stamps.txt 2>&1; cat $O/stamps.txt
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
