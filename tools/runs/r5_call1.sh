#!/bin/bash
# round 5, call 1: the rebuilt probe kernel + DPP scans + pre-bound step: GPU suite, bench line, A/B of the tile shapes, phase
# stamps before / after, kernel trace, 16x PMC "before"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c1; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py tests/test_module_gpu.py tests/test_demb_gpu.py -q -m gpu -x > $O/pytest_a.txt 2>&1; tail -5 $O/pytest_a.txt
timeout 300 python tools/ab_probe_c.py > $O/ab_probe_c.txt 2>&1; tail -8 $O/ab_probe_c.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5c1/bench.json').read().strip().splitlines()[-1])
    print('ms', d['ms_per_step'], 'sus', d['sustained']['ms_per_step'], 'autograd', d.get('step_via_autograd_ms'), 'frac', d['roofline']['frac'], 'step', d['step_roofline']['frac'], '16x', d['c2_16x']['ms_per_step'], d['c2_16x']['step_roofline']['frac'])
    print({k: v['ms'] for k, v in d['roofline']['kernels'].items()})
except Exception as e: print('bench parse failed', e)
PY
# phase stamps: old kernel, new kernel
for V in 0 1; do
  MI355_PROBE_C=$V MI355_LIB=$R/recsys-examples_amd/lib/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py > $O/stamps_v$V.txt 2>&1
done
head -30 $O/stamps_v1.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c2_kernel_trace_stats.txt; head -12 $O/c2_kernel_trace_stats.txt
python $R/tools/rocpd_timeline.py $DB 40 > $O/step_timeline.txt
rm -rf /tmp/kt
# 16x: kernel trace + PMC traffic
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt16 -o t -- python $R/tools/step_16x.py > $O/step16.log 2>&1
DB=$(find /tmp/kt16 -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c2_16x_kernel_trace_stats.txt; head -14 $O/c2_16x_kernel_trace_stats.txt
rm -rf /tmp/kt16
: > $O/pmc_16x.txt
i=0
for SET in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  timeout 400 rocprofv3 --pmc $SET -d /tmp/pmc$i -o p -- python $R/tools/step_16x.py --steps 3 > /tmp/pmc$i.log 2>&1
  DB=$(ls /tmp/pmc$i/*/*.db /tmp/pmc$i/*.db 2>/dev/null | head -1)
  echo "## pass $i: $SET" >> $O/pmc_16x.txt
  python $R/tools/pmc_dump.py $DB mi355 >> $O/pmc_16x.txt 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_16x.txt $O/pmc_traffic_16x.json
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_fused_fwd_gpu.py --deselect tests/test_module_gpu.py --deselect tests/test_demb_gpu.py > $O/pytest_b.txt 2>&1; tail -5 $O/pytest_b.txt
