#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu -k "exchange or golden or random_jagged" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
( for v in "" _fb2 _fb4; do echo "== lib$v"; MI355_LIB=$L/librecsys_amd$v.so timeout 300 python tools/hstu_shapes.py --seeds 1 | head -4; done ) > $O/shapes.txt 2>&1; grep -v amdgpu.ids $O/shapes.txt
MI355_LIB=$L/librecsys_amd_tim.so python tools/hstu_phase_cycles.py --bwdpc --batch 8 --seqlen 4096 2>&1 | grep -v amdgpu
