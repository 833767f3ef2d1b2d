#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
for cf in 0 1.25; do
timeout 600 python bench.py --force-sharded --no-hstu --no-cpu-baseline --capacity-factor $cf > $O/sharded_w1_cf$cf.json 2> $O/err.txt
python - <<PY
import json
d=json.loads(open('gpurun_out/r4o/sharded_w1_cf$cf.json').read().strip().splitlines()[-1])
print('cf $cf ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], {k: round(v,4) if isinstance(v,float) else v for k,v in d['stages_ms'].items() if k!='note'})
PY
done
