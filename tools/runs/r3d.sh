#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_demb_gpu.py tests/test_module_gpu.py tests/test_fused_fwd_gpu.py tests/test_twin_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-hstu --steps 100 --warmup 10"
$B > $O/bench_flat.json 2> $O/err.txt
MI355_POOL_VARIANT=30 $B > $O/bench_pipe.json 2>> $O/err.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d['ms_per_step']*1e3,1), round(d['sustained']['ms_per_step']*1e3,1), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
