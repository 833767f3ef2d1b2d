#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3l; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd $R
for v in stamps st_noevict st_nocsr; do
  echo "=== $v"
  MI355_LIB=$L/librecsys_amd_$v.so timeout 300 python tools/index_phase_stamps.py > $O/stamps_$v.txt 2>&1
  grep -A3 "part2 blocks" $O/stamps_$v.txt | head -2; grep "block life\|wall clock" $O/stamps_$v.txt | sed -n 3,4p
  sed -n '/absolute wall/,$p' $O/stamps_$v.txt
done
