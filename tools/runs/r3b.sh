#!/bin/bash
# round-3 GPU run b: row-traffic microbench, gather KIT / UNR variants, side-stream overlap of the index kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3b; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/tools/ubench_rows.bin > $O/ubench.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --no-hstu --steps 100 --warmup 10"
$B > $O/bench_base.json 2> $O/err.txt
for v in kit1 kit2 kit3 kit6; do MI355_LIB=$L/librecsys_amd_$v.so $B > $O/bench_$v.json 2>> $O/err.txt; done
MI355_POOL_VARIANT=21 $B > $O/bench_unr8.json 2>> $O/err.txt
MI355_FUSED_SIDE=1 $B > $O/bench_side.json 2>> $O/err.txt
MI355_FUSED_SIDE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/side_stats.txt 2>&1
python $R/tools/rocpd_timeline.py $DB 60 > $O/side_timeline.txt 2>&1
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d['ms_per_step']*1e3,1), round(d['sustained']['ms_per_step']*1e3,1), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
