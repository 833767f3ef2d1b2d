#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3f; mkdir -p $O
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
tail -3 $O/err.txt
