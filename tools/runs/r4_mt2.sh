#!/bin/bash
# round 4: kernel traces of the model shapes (tools/bench_model_shapes.py --only N): args = case numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4mt; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace$c -o t -- python $R/tools/bench_model_shapes.py --steps 50 --only $c > $O/trace$c.log 2>&1
  echo "== case $c"; python $R/tools/rocpd_stats.py $O/trace$c/t_results.db | head -8 | cut -c1-150
done
