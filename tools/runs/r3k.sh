#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3k; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd $R
MI355_LIB=$L/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py > $O/stamps.txt 2>&1
sed -n '/absolute wall/,$p' $O/stamps.txt
MI355_FUSED_PART=1 MI355_LIB=$L/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py > $O/stamps_a.txt 2>&1
sed -n '/absolute wall/,$p' $O/stamps_a.txt
