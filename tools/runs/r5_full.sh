#!/bin/bash
# round 5: full GPU suite + default bench line + smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5full; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; grep "passed\|failed" $O/pytest_gpu.txt; grep "hstu_tolerance_used" $O/pytest_gpu.txt | head -12
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5full/bench.json').read().strip().splitlines()[-1])
    print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'impl', d.get('step_via_impl_ms'), 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'])
    print('hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], 'jag', d['hstu_jagged']['fwd_ms'], d['hstu_jagged']['bwd_ms'])
    c=d['cpu_baseline']; print('cpu', c['value'], c['cores'], c['value_by_threads'])
    print({k: (v.get('ms_per_step'), v.get('step_roofline', {}).get('frac')) for k, v in d['model_shapes'].items() if isinstance(v, dict)})
except Exception as e: print('bench parse failed', e)
PY
timeout 300 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
