#!/bin/bash
# round 4: S-wave / K-wave dK pass of the attention backward
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -8 $O/tests.txt
( echo "== kvpc (default)"; timeout 300 python tools/hstu_shapes.py --seeds 1
  echo "== 4-wave dK pass"; MI355_HSTU_KVPC=0 timeout 300 python tools/hstu_shapes.py --seeds 1 ) > $O/shapes.txt 2>&1; grep -v amdgpu.ids $O/shapes.txt
