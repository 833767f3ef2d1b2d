#!/bin/bash
# round 5, call 5: the big-batch stage (tile-major records + split + partition kernel over 4 096-record lists)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c5; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1200 python -m pytest tests/test_fused_fwd_gpu.py tests/test_twin_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -4 $O/pytest_a.txt
timeout 300 python tools/step_16x.py --steps 6 > $O/step16.txt 2>&1; tail -2 $O/step16.txt
MI355_BIG=0 timeout 300 python tools/step_16x.py --steps 6 > $O/step16_b.txt 2>&1; tail -1 $O/step16_b.txt
timeout 300 python tools/step_16x.py --steps 6 --mult 4 > $O/step4.txt 2>&1; tail -1 $O/step4.txt
MI355_BIG=0 timeout 300 python tools/step_16x.py --steps 6 --mult 4 > $O/step4_b.txt 2>&1; tail -1 $O/step4_b.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt16 -o t -- python $R/tools/step_16x.py > $O/step16_trace.log 2>&1
DB=$(find /tmp/kt16 -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c2_16x_kernel_trace_stats.txt; head -10 $O/c2_16x_kernel_trace_stats.txt
cd $R
timeout 600 python bench.py --no-cpu-baseline --no-hstu > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5c5/bench.json').read().strip().splitlines()[-1])
    print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'impl', d.get('step_via_impl_ms'), 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'])
except Exception as e: print('bench parse failed', e)
PY
