#!/bin/bash
# round 5, call 24: ms/step against --steps
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c24; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
for st in 50 300 50 100 200; do
timeout 300 python bench.py --steps $st --warmup 10 --no-cpu-baseline --no-hstu --no-extra --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $st ms', round(d['ms_per_step'],5), 'sus', round(d['sustained']['ms_per_step'],5), 'impl', d.get('step_via_impl_ms'))"
done
