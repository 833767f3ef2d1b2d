#!/bin/bash
# round 4: kernel traces of the 8-table pooled shape (path (c), table-aligned partitions) and of C2 through the same tool
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4mt; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in 3 0 2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace$c -o t -- python $R/tools/bench_model_shapes.py --steps 50 --only $c > $O/trace$c.log 2>&1
  f=$(ls $O/trace$c/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 $f | cut -c1-150
done
