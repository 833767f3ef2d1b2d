#!/bin/bash
# round 5, call 28: phase stamps of the index kernels in the eviction regime (300 distinct batches first)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c28; mkdir -p $O; cd $R
MI355_LIB=$R/recsys-examples_amd/lib/librecsys_amd_stamps.so timeout 600 python tools/index_phase_stamps.py --batches 300 > $O/stamps_fill300.txt 2>&1
grep -v "^/opt" $O/stamps_fill300.txt | head -60 | cut -c1-150
