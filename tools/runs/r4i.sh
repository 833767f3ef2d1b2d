#!/bin/bash
# round 4, call 9: 8-wave one-GEMM passes of the attention backward (dV from P, dQ from dS)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
L=$R/recsys-examples_amd/lib
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -8 $O/tests.txt
( echo "== x8 (default)"; timeout 300 python tools/hstu_shapes.py --seeds 1
  echo "== x8 VBUF=3"; MI355_LIB=$L/librecsys_amd_x8v3.so timeout 300 python tools/hstu_shapes.py --seeds 1
  echo "== 4-wave passes"; MI355_HSTU_X8=0 timeout 300 python tools/hstu_shapes.py --seeds 1 ) > $O/shapes.txt 2>&1; grep -v amdgpu.ids $O/shapes.txt
