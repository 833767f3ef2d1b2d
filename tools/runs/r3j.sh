#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3j; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd $R
( time timeout 600 python -m pytest tests/test_fused_fwd_gpu.py tests/test_module_gpu.py tests/test_twin_gpu.py tests/test_demb_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-hstu --steps 100 --warmup 10"
$B > $O/bench_c.json 2> $O/err.txt
MI355_FUSED_PART=1 $B > $O/bench_a.json 2>> $O/err.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d['ms_per_step']*1e3,1), round(d['sustained']['ms_per_step']*1e3,1), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
MI355_LIB=$L/librecsys_amd_stamps.so timeout 300 python tools/index_phase_stamps.py 2>&1 | cat > $O/stamps.txt
cat $O/stamps.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB | head -6 | cut -c1-150 | tee $O/stats.txt
