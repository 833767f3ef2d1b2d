#!/bin/bash
# round 5, call 15: the exchanges of the sharded step inside the library (csrc/exchange.hip) -- tests, forced W = 1 bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c15; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_sharded_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B8 "Error\|assert " $O/pytest_a.txt | head -60
for v in 1 0; do
MI355_NATIVE_EXCHANGE=$v timeout 600 python bench.py --force-sharded --no-hstu --no-cpu-baseline --no-extra > $O/sharded_w1_native$v.json 2> $O/sharded_w1_native$v.err; tail -3 $O/sharded_w1_native$v.err
python - <<PY
import json
try:
    d=json.loads(open('$O/sharded_w1_native$v.json').read().strip().splitlines()[-1])
    print('native=$v ms', d['ms_per_step'], {k: (round(x,4) if isinstance(x,float) else x) for k,x in d.get('stages_ms',{}).items() if k!='note'})
except Exception as e: print('parse failed', e)
PY
done
timeout 300 python tools/runs/prof_sharded_host.py > $O/prof_host.txt 2>&1; head -60 $O/prof_host.txt
