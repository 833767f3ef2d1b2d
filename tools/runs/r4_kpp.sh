#!/bin/bash
# round 4: keys per partition of path (c) for batches below 256 K keys (sequence shapes), A/B in one call
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4mt; mkdir -p $O; cd $R
for k in 1024 512 256; do
  for c in 1 2 4; do MI355_FUSED_KPP=$k timeout 200 python tools/bench_model_shapes.py --steps 100 --only $c 2>&1 | grep -v amdgpu | sed "s/^/kpp $k: /" | cut -c1-150; done
done | tee $O/kpp_ab.txt
