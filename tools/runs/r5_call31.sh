#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c31; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
for v in 1 2 3; do timeout 120 python tools/runs/sharded_w1_loop.py 2>&1 | grep "ms/step\|Error"; done
timeout 900 python -m pytest tests/test_sharded_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt
timeout 600 python bench.py --force-sharded --no-hstu --no-cpu-baseline --no-extra > $O/sharded_w1.json 2> /dev/null
python - <<PY
import json
d=json.loads(open('$O/sharded_w1.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], {k: (round(x,4) if isinstance(x,float) else x) for k,x in d.get('stages_ms',{}).items() if k!='note'})
PY
