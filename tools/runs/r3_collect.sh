#!/bin/bash
# round-3 evidence in one box call: tools/collect_profiles.sh r03 + phase stamps + the micro-benchmarks of DESIGN.md section 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; L=$R/recsys-examples_amd/lib
cd $R
bash tools/collect_profiles.sh r03
MI355_LIB=$L/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py > $O/r03_index_phase_stamps.txt 2>&1
timeout 200 ./tools/ubench_rows.bin > $O/r03_ubench_rows.txt 2>&1
timeout 200 ./tools/ubench_gather.bin > $O/r03_ubench_gather.txt 2>&1
timeout 100 ./tools/ubench_boundary.bin > $O/r03_ubench_boundary.txt 2>&1
ls -la $O/r03_*
