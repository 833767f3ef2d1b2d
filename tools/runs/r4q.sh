#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
( MI355_HSTU_PAIR=0 timeout 120 python tools/hstu_fwd_ab.py --shapes c3,d4096,d8x4096,jag1,jag2,ragged,d1024
  timeout 120 python tools/hstu_fwd_ab.py --shapes c3,d4096,d8x4096,jag1,jag2,ragged,d1024 ) > $O/ab.txt 2>&1
grep -v amdgpu.ids $O/ab.txt
timeout 900 python -m pytest tests/test_hstu_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
