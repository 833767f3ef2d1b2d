#!/bin/bash
# round 5, call 38: the backward's hot-row thresholds (MI355_HOT / MI355_WAVE) swept on the C2 step (bench.py kernel timing)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in "4 128" "2 128" "8 128" "4 64" "4 256" "3 128" "6 192" "4 128"; do
  set -- $cfg
  MI355_HOT=$1 MI355_WAVE=$2 timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hstu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HOT $1 WAVE $2: ms', round(d['ms_per_step'],5), 'bwd_kernel us', round(d['roofline']['kernels']['bwd_kernel']['ms']*1e3,2))"
done
