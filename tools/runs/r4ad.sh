#!/bin/bash
# the two-waves-per-SIMD kernels without AGPR pins (all 256 registers architectural, no v_accvgpr copies) against the pinned build
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4ad; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu -k "not kernel_variants" 2>&1 | tail -4
( for v in _ag1 "" _ag1 ""; do echo "== lib$v"; MI355_LIB=$L/librecsys_amd$v.so timeout 300 python tools/hstu_shapes.py --seeds 1 2>&1 | grep "C3\|x 4096\|seed 1\|<= 512\|uniform"; done ) > $O/agpr.txt 2>&1; cat $O/agpr.txt
( MI355_HSTU_DS_MAX_BYTES=1073741824 MI355_LIB=$L/librecsys_amd_tim.so timeout 200 python tools/hstu_phase_cycles.py --bwdpc --batch 8 --seqlen 4096 2>&1 | grep -v amdgpu.ids
  MI355_HSTU_PAIR=0 MI355_LIB=$L/librecsys_amd_tim.so timeout 200 python tools/hstu_phase_cycles.py --q2 --batch 8 --seqlen 4096 2>&1 | grep -v amdgpu.ids ) > $O/stamps.txt 2>&1; cat $O/stamps.txt
