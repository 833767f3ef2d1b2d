#!/bin/bash
# round 5, call 12: cell layout forced at C2 (MI355_CELLS=2) against the list layout; MT test again
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c12; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B5 "Error\|assert" $O/pytest_a.txt | head -40
timeout 300 python tools/ab_probe_c.py --var MI355_CELLS --variants 0,2 > $O/ab_cells.txt 2>&1; tail -4 $O/ab_cells.txt
cd /tmp && export TMPDIR=/tmp
for c in 2; do
MI355_CELLS=$c timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt$c -o t -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt$c.log 2>&1
DB=$(find /tmp/kt$c -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c2_kernel_trace_stats_cells$c.txt; head -6 $O/c2_kernel_trace_stats_cells$c.txt
tail -c 600 /tmp/kt$c.log
done
