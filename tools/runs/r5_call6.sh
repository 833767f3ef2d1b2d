#!/bin/bash
# round 5, call 6: big-batch stage with few large partitions + streaming partition kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c6; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -x -k "big_batches or every_tile_shape" > $O/pytest_a.txt 2>&1; tail -4 $O/pytest_a.txt
timeout 300 python tools/step_16x.py --steps 6 > $O/step16.txt 2>&1; tail -1 $O/step16.txt
timeout 300 python tools/step_16x.py --steps 6 --mult 4 > $O/step4.txt 2>&1; tail -1 $O/step4.txt
timeout 300 python tools/step_16x.py --steps 6 --mult 8 > $O/step8.txt 2>&1; tail -1 $O/step8.txt
MI355_BIG=0 timeout 300 python tools/step_16x.py --steps 6 --mult 8 > $O/step8_b.txt 2>&1; tail -1 $O/step8_b.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt16 -o t -- python $R/tools/step_16x.py > $O/step16_trace.log 2>&1
DB=$(find /tmp/kt16 -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c2_16x_kernel_trace_stats.txt; head -7 $O/c2_16x_kernel_trace_stats.txt
