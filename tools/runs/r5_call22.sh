#!/bin/bash
# round 5, call 22: overflow re-run chain on the side stream (mode 2) -- flood tests, C2 step A/B of modes 0 / 1 / 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c22; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py -q -m gpu -x -k "rerun" > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B12 "Error\|assert " $O/pytest_a.txt | head -70
for rep in 1 2 3; do for m in 0 2 1; do
MI355_FUSED_OVERFLOW_RERUN=$m timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-hstu --no-extra --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rerun=$m ms', round(d['ms_per_step'],5), 'sus', round(d['sustained']['ms_per_step'],5), 'seq', d.get('model_shapes',{}).get('sequence_8x16384_tokens',{}).get('ms_per_step'))"
done; done
