#!/bin/bash
# round 4, last call: full GPU suite + the bench line + the rocprofv3 kernel-trace summary of the same bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4final; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /tmp/kt.log 2>&1
  DB=$(find /tmp/kt -name '*.db' | head -1); python $R/tools/rocpd_stats.py $DB > $O/c2_kernel_trace_stats.txt; head -8 $O/c2_kernel_trace_stats.txt | cut -c1-150 )
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4final/bench.json').read().strip().splitlines()[-1])
print('shapes', {k: (round(v['ms_per_step'], 4), round(v['eval_forward_ms'], 4), v['path_c']) for k, v in d['model_shapes'].items() if k != 'config'})
print('ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['step_roofline']['frac'], 'hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'])
PY
