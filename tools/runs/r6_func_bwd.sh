#!/bin/bash
# round 6: `func` masks on the exchange backward: parity tests, the mask-function bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_hstu_gpu.py -q -x 2>&1 | tail -5
{ python tools/bench_hstu_func.py | grep "func in"
  python tools/bench_hstu_func.py --batch 32 --seqlen 512 | grep "func in"
  MI355_HSTU_DS_MAX_BYTES=0 python tools/bench_hstu_func.py | grep "func in"
  python tools/bench_hstu.py --batch 8 --seqlen 4096 --reps 10 2>&1 | tail -1
  python tools/bench_hstu.py --batch 32 --seqlen 512 --reps 10 2>&1 | tail -1
} > $O/r06_hstu_func_bwd.txt 2>&1
cat $O/r06_hstu_func_bwd.txt
