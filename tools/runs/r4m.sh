#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 600 python bench.py --force-sharded --no-hstu --no-cpu-baseline > $O/sharded_w1.json 2> $O/sharded_w1.err; tail -c 300 $O/sharded_w1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4m/sharded_w1.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'ranks', d['ranks'], 'stages', d['stages_ms'])
PY
