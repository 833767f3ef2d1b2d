#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3r; mkdir -p $O
cd $R
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-hstu --no-extra --steps 100 --warmup 10"
$B > $O/bench_base.json 2> $O/err.txt
for hw in "2 64" "6 64" "8 64" "4 32" "4 128" "4 256"; do set -- $hw; MI355_HOT=$1 MI355_WAVE=$2 $B > $O/bench_hot$1_wave$2.json 2>> $O/err.txt; done
MI355_CHUNK=512 $B > $O/bench_chunk512.json 2>> $O/err.txt
MI355_CHUNK=2048 $B > $O/bench_chunk2048.json 2>> $O/err.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d['ms_per_step']*1e3,1), round(d['sustained']['ms_per_step']*1e3,1), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
