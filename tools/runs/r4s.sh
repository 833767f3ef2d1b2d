#!/bin/bash
# dK pass with a second (phase-aligning) barrier per step: A/B + stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
( for v in "" _mb1 "" _mb1; do echo "== lib$v"; MI355_LIB=$L/librecsys_amd$v.so timeout 300 python tools/hstu_shapes.py --seeds 1 2>&1 | grep "C3\|x 4096\|seed 1"; done ) > $O/shapes.txt 2>&1; cat $O/shapes.txt
( MI355_LIB=$L/librecsys_amd_tmb1.so timeout 200 python tools/hstu_phase_cycles.py --bwdpc --batch 8 --seqlen 4096 ) > $O/stamps.txt 2>&1; cat $O/stamps.txt
