#!/bin/bash
# forced-sharded W = 1 step after the stream-query cleanup: timing, kernel timeline, host profile
mkdir -p gpurun_out/r3s2
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for m in partial rows; do
  timeout 300 python bench.py --force-sharded --shard-mode $m --steps 200 --warmup 20 --no-hstu --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/r3s2/bench_$m.json 2> gpurun_out/r3s2/bench_$m.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/r3s2/bench_$m.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'])"
done
timeout 300 python -c "
import cProfile, pstats, sys, runpy
sys.argv=['bench.py','--force-sharded','--shard-mode','partial','--steps','800','--warmup','20','--no-hstu','--no-cpu-baseline','--no-kernel-timing','--no-extra']
cProfile.run('runpy.run_path(\"bench.py\", run_name=\"__main__\")', '/tmp/prof.out')
" > /dev/null 2> gpurun_out/r3s2/cprof.err
python -c "
import pstats; p=pstats.Stats('/tmp/prof.out'); p.sort_stats('cumulative').print_stats(60)" > gpurun_out/r3s2/cprofile.txt 2>&1
python -c "
import pstats; p=pstats.Stats('/tmp/prof.out'); p.sort_stats('tottime').print_stats(40)" > gpurun_out/r3s2/cprofile_tottime.txt 2>&1
bash tools/prof_sharded.sh
mv gpurun_out/sh_* gpurun_out/r3s2/ 2>/dev/null
