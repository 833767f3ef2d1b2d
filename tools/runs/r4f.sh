#!/bin/bash
# round 4, call 6: full GPU suite + bench line after the round's first batch of fixes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'autograd', d.get('step_via_autograd_ms'), 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['step_roofline']['frac'])
print('hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], 'jag', d['hstu_jagged']['fwd_ms'], d['hstu_jagged']['bwd_ms'])
c=d['cpu_baseline']; print('cpu', c['value'], c['cores'], c['value_by_threads'], c['physical_cores'], c['index_add_port'], c['c1']['value_by_threads'])
PY
