#!/bin/bash
# round 5, call 9: HSTU backward bookkeeping (plan kernel from LDS, one chunk when the bound fits) + full suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c9; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5c9/bench.json').read().strip().splitlines()[-1])
    print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'impl', d.get('step_via_impl_ms'), 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'])
    print('hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], 'jag', d['hstu_jagged']['fwd_ms'], d['hstu_jagged']['bwd_ms'], 'l4096', {k: v for k, v in d['hstu_l4096'].items() if 'TFLOP' in k or 'ms' in k})
except Exception as e: print('bench parse failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kth -o t -- python $R/tools/bench_hstu.py > $O/hstu_trace.log 2>&1
DB=$(find /tmp/kth -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/hstu_kernel_trace_stats.txt; head -12 $O/hstu_kernel_trace_stats.txt
