#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3h; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd $R
( time timeout 600 python -m pytest tests/test_fused_fwd_gpu.py tests/test_module_gpu.py tests/test_twin_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-hstu --steps 100 --warmup 10"
$B > $O/bench_kit7.json 2> $O/err.txt
for k in 4 5 6; do MI355_LIB=$L/librecsys_amd_p2kit$k.so $B > $O/bench_kit$k.json 2>> $O/err.txt; done
MI355_FUSED_PART=1 $B > $O/bench_patha.json 2>> $O/err.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d['ms_per_step']*1e3,1), round(d['sustained']['ms_per_step']*1e3,1), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
