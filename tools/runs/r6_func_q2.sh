#!/bin/bash
# round 6: `func` masks on the 64-rows-per-wave forward: parity tests, the mask-function bench (new default | MI355_HSTU_FWD=5 = the one-kind kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_hstu_gpu.py -q -x -k "mask or func or arbitrary" 2>&1 | tail -5
{ for hook in 0 5; do
    echo "== MI355_HSTU_FWD=$hook"
    MI355_HSTU_FWD=$hook python tools/bench_hstu_func.py | grep "func in"
    MI355_HSTU_FWD=$hook python tools/bench_hstu_func.py --batch 32 --seqlen 512 | grep "func in"
  done
  python tools/bench_hstu.py --batch 8 --seqlen 4096 --reps 10 2>&1 | tail -2
  python tools/bench_hstu.py --batch 32 --seqlen 512 --reps 10 2>&1 | tail -2
} > $O/r06_hstu_func_q2.txt 2>&1
cat $O/r06_hstu_func_q2.txt
