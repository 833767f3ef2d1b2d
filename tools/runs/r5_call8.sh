#!/bin/bash
# round 5, call 8: streaming partition kernel with the keys in the records' output entries; 1 024-key stage tiles (A/B)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c8; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -x -k "big_batches" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
for T in 2048 1024; do
  MI355_BIG_TILE=$T timeout 300 python tools/step_16x.py --steps 6 > $O/step16_$T.txt 2>&1; tail -1 $O/step16_$T.txt
done
MI355_BIG_TILE=1024 timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -x -k "big_batches" > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
MI355_LIB=$R/recsys-examples_amd/lib/librecsys_amd_stamps.so timeout 300 python tools/stamps_16x.py 16 > $O/stamps16.txt 2>&1; tail -12 $O/stamps16.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt16 -o t -- python $R/tools/step_16x.py > $O/step16_trace.log 2>&1
DB=$(find /tmp/kt16 -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c2_16x_kernel_trace_stats.txt; head -7 $O/c2_16x_kernel_trace_stats.txt
