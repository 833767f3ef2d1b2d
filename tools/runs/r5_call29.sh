#!/bin/bash
# round 5, call 29: parallel backward plan kernel -- attention tests, plan kernel time in a default bench run, jagged backward time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c29; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 1200 python -m pytest tests/test_hstu_gpu.py tests/test_full_size_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; grep "passed\|failed" $O/pytest_a.txt; grep -B12 "Error\|assert " $O/pytest_a.txt | head -50
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kj -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > /tmp/kj.log 2>&1
db=$(find /tmp/kj -name '*.db' | head -1)
{ echo "== kernels of a whole default bench.py run (C2 step, 16x step, model shapes, attention incl. the jagged C4 shape: hstu_bwd_plan_kernel)"; python $R/tools/rocpd_stats.py $db | grep -i "kernel \|hstu\|total" | cut -c1-160; } > $O/bench_attention_kernels.txt 2>&1
grep -i "plan\|kernel  " $O/bench_attention_kernels.txt | cut -c1-140
python - <<'PY'
import json
d=json.loads([l for l in open("/tmp/kj.log") if l.startswith("{")][-1]); print('jagged', d['hstu_jagged']['fwd_ms'], d['hstu_jagged']['bwd_ms'], 'hstu', d['hstu']['fwd_ms'], d['hstu']['bwd_ms'])
PY
