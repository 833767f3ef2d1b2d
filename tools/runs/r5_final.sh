#!/bin/bash
# round 5, final tree: full GPU suite, smoke, default bench line (-> profiles/r05_c2_bench.json), kernel trace of the same command
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1800 python -m pytest tests -q -m gpu > $O/r05_pytest_gpu.txt 2>&1; grep "passed\|failed" $O/r05_pytest_gpu.txt
timeout 300 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke.log 2>&1; tail -1 $O/r05_smoke.log
cd /tmp && export TMPDIR=/tmp
( time timeout 900 python $R/bench.py ) > $O/r05_c2_bench.json 2> $O/r05_c2_bench.err; tail -4 $O/r05_c2_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/r05_c2_kernel_trace_stats.txt
python $R/tools/rocpd_timeline.py $DB 80 > $O/r05_step_timeline.txt
head -6 $O/r05_c2_kernel_trace_stats.txt | cut -c1-60,72-140
cd $R
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c2_bench.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'])
print('filling', d['c2_table_filling']['ms_per_step'], 'hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], 'jag', d['hstu_jagged']['fwd_ms'], d['hstu_jagged']['bwd_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
