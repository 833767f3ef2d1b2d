#!/bin/bash
# round-6 evidence in one box call: tools/collect_profiles.sh r06 (bench line, kernel trace + timeline, PMC passes + traffic JSON,
# attention SQ counters, extended configurations, smoke) + forced-sharded W = 1 bench, attention kernel traces and L2-miss traffic,
# the pipelined loop, the mask-function bench, full GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd $R
export MASTER_ADDR=127.0.0.1
bash tools/collect_profiles.sh r06
timeout 600 python bench.py --force-sharded --no-hstu --no-cpu-baseline --no-extra > $O/r06_sharded_w1_bench.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp
  for cfg in "c3 32 512" "l4096 8 4096"; do
    set -- $cfg; rm -rf /tmp/prof_$1
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o t -- python $R/tools/bench_hstu.py --batch $2 --seqlen $3 --reps 10 > /dev/null 2>&1
    db=$(find /tmp/prof_$1 -name '*.db' | head -1)
    echo "== attention kernels, batch $2 x L $3 (rocprofv3 --kernel-trace)"; python $R/tools/rocpd_stats.py $db | grep -i "kernel \|hstu\|total" | cut -c1-160
  done > $O/r06_hstu_kernel_trace_stats.txt 2>&1
)
timeout 900 bash tools/pmc_hstu_traffic.sh 8 4096 gpurun_out/r06_pmc_hstu_traffic.txt > /dev/null 2>&1
timeout 300 python tools/pipeline_step.py > $O/r06_pipeline_step.txt 2>&1
{ python tools/bench_hstu_func.py; python tools/bench_hstu_func.py --batch 32 --seqlen 512; MI355_HSTU_DS_MAX_BYTES=0 MI355_HSTU_FWD=5 python tools/bench_hstu_func.py; } 2>&1 | grep "func in" > $O/r06_hstu_func_final.txt
timeout 300 python tools/bench_model_shapes.py > $O/r06_model_shapes.txt 2>&1
timeout 1800 python -m pytest tests -q -m gpu > $O/r06_pytest_gpu.txt 2>&1; grep "passed\|failed" $O/r06_pytest_gpu.txt
ls -la $O/r06_* | awk '{print $5, $9}'
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_c2_bench.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'], 'pipelined', d.get('pipelined_ms_per_step'))
print('hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], 'l4096', d['hstu_l4096']['fwd_ms'], d['hstu_l4096']['bwd_ms'])
d=json.loads(open('gpurun_out/r06_sharded_w1_bench.json').read().strip().splitlines()[-1]); print('sharded w1', d['ms_per_step'], d['exchange'], d['stages_ms'])
PY
