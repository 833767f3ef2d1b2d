#!/bin/bash
# round 5, call 23: why 0.16 ms?  default bench line + kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c23; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-hstu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', round(d['ms_per_step'],5), 'sus', round(d['sustained']['ms_per_step'],5), {k: round(v['ms'],4) for k,v in d['roofline']['kernels'].items()})"
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-hstu --no-extra --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-kernel-timing: ms', round(d['ms_per_step'],5), 'sus', round(d['sustained']['ms_per_step'],5))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kt.txt; head -8 $O/kt.txt | cut -c1-60,72-140
python $R/tools/rocpd_timeline.py $DB 12 | cut -c1-110
