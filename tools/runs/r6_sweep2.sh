#!/bin/bash
# round 6: second micro-sweep of the row movers (bags per lane group of the gather at 2 rows per round; row groups per lane group of the small-batch backward)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd $R
{ for rep in 1 2; do for v in "" _bk1 _bk3 _gk2 _gk6 _gk8; do
    L=$R/recsys-examples_amd/lib/librecsys_amd$v.so
    MI355_LIB=$L python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-hstu --no-extra 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('variant %-8s step %.4f sustained %.4f  ' % ('${v:-default}', d['ms_per_step'], d.get('sustained',{}).get('ms_per_step',0)), {n: round(v['ms']*1e3,1) for n,v in k.items()})
"
  done; done
} > $O/r06_sweep2.txt 2>&1
cat $O/r06_sweep2.txt
