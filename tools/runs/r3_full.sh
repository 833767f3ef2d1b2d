#!/bin/bash
# full GPU suite + the round-3 evidence collection + HSTU shapes, one box call
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd $R
export MASTER_ADDR=127.0.0.1
timeout 2400 python -m pytest tests -q -m gpu -x > $O/r03_pytest_gpu.txt 2>&1
tail -3 $O/r03_pytest_gpu.txt
bash tools/runs/r3_collect.sh > $O/r03_collect.log 2>&1
timeout 300 python tools/hstu_shapes.py --seeds 4 > $O/r03_hstu_shapes.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke.log 2>&1; tail -1 $O/r03_smoke.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_c2_bench.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['step_roofline']['frac'])
print('hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], 'jag', d['hstu_jagged']['fwd_ms'], d['hstu_jagged']['bwd_ms'])
PY
