#!/bin/bash
# round 5, call 10: partition blocks in the gather's launch (A/B), VMM destroy behind a device sync, suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c10; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 300 python tools/ab_probe_c.py --var MI355_PART_FUSED --variants 0,1,2 > $O/ab_part_fused.txt 2>&1; tail -5 $O/ab_part_fused.txt
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py tests/test_twin_gpu.py tests/test_module_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt
timeout 600 python bench.py --no-cpu-baseline --no-hstu > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5c10/bench.json').read().strip().splitlines()[-1])
    print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'impl', d.get('step_via_impl_ms'), 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'])
    print({k: v['ms'] for k, v in d['roofline']['kernels'].items()})
    print({k: (v.get('ms_per_step'), v.get('step_roofline', {}).get('frac')) for k, v in d['model_shapes'].items() if isinstance(v, dict)})
except Exception as e: print('bench parse failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c2_kernel_trace_stats.txt; head -6 $O/c2_kernel_trace_stats.txt
