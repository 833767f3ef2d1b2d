#!/bin/bash
# round-5 evidence in one box call: tools/collect_profiles.sh r05 + phase stamps, sharded W = 1 (bench / timeline / host profile),
# attention kernel traces (incl. the jagged shape's plan kernel), model shapes, full GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; L=$R/recsys-examples_amd/lib
cd $R
export MASTER_ADDR=127.0.0.1
bash tools/collect_profiles.sh r05
MI355_LIB=$L/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py > $O/r05_index_phase_stamps_after.txt 2>&1
timeout 600 python bench.py --force-sharded --no-hstu --no-cpu-baseline --no-extra > $O/r05_sharded_w1_bench.json 2> /dev/null
timeout 300 python tools/runs/prof_sharded_host.py > $O/r05_sharded_w1_host_profile.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace -d /tmp/sh -o t -- python $R/tools/runs/sharded_w1_loop.py > /tmp/sh.log 2>&1
  db=$(find /tmp/sh -name '*.db' | head -1)
  { grep "ms/step" /tmp/sh.log; python $R/tools/rocpd_timeline.py $db 60; } > $O/r05_sharded_w1_timeline.txt
  python $R/tools/rocpd_stats.py $db > $O/r05_sharded_w1_stats.txt
  for cfg in "c3 32 512" "l4096 8 4096"; do
    set -- $cfg; rm -rf /tmp/prof_$1
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o t -- python $R/tools/bench_hstu.py --batch $2 --seqlen $3 --reps 10 > /dev/null 2>&1
    db=$(find /tmp/prof_$1 -name '*.db' | head -1)
    echo "== attention kernels, batch $2 x L $3 (rocprofv3 --kernel-trace)"; python $R/tools/rocpd_stats.py $db | grep -i "kernel \|hstu\|total" | cut -c1-160
  done > $O/r05_hstu_kernel_trace_stats.txt 2>&1
  rm -rf /tmp/kj
  rocprofv3 --kernel-trace --stats -d /tmp/kj -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > /tmp/kj.log 2>&1
  db=$(find /tmp/kj -name '*.db' | head -1)
  { echo "== kernels of a whole default bench.py run (C2 step, 16x step, model shapes, attention incl. the jagged C4 shape: hstu_bwd_plan_kernel)"; python $R/tools/rocpd_stats.py $db | grep -i "kernel \|hstu\|total" | cut -c1-160; } > $O/r05_bench_attention_kernels.txt 2>&1
)
timeout 300 python tools/bench_model_shapes.py > $O/r05_model_shapes.txt 2>&1
timeout 1800 python -m pytest tests -q -m gpu > $O/r05_pytest_gpu.txt 2>&1; grep "passed\|failed" $O/r05_pytest_gpu.txt
ls -la $O/r05_* | awk '{print $5, $9}'
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c2_bench.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'])
d=json.loads(open('gpurun_out/r05_sharded_w1_bench.json').read().strip().splitlines()[-1]); print('sharded w1', d['ms_per_step'], d['stages_ms'])
PY
