#!/bin/bash
# round 5, call 25: the step at 300+ batches (table filling up): which kernel grows?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5c25; mkdir -p $O; cd $R
export MASTER_ADDR=127.0.0.1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kt300.txt; head -8 $O/kt300.txt | cut -c1-60,72-140
python $R/tools/rocpd_timeline.py $DB 16 | cut -c1-110
tail -c 600 /tmp/kt.log | grep -o '"ms_per_step": [0-9.]*' | head -2
cd $R
python - <<'PY'
import sys, os, torch
sys.path.insert(0, 'recsys-examples_amd'); sys.path.insert(0, '.')
import bench
dev = torch.device('cuda')
batches = bench.zipf_batches(10_000_000, 0.99, 65536, 320, dev)
m = bench.build_module(10_000_000, 128, dev); m.train()
grad = (torch.randn(65536, 128, device=dev) * 0.01).to(torch.bfloat16)
import time
for i, (k, o) in enumerate(batches):
    out, st = m._forward_impl(k, o, train=True); m._backward_impl(st, grad)
    if i % 40 == 39:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k2, o2 in batches[i-39:i+1]:
            out, st = m._forward_impl(k2, o2, train=True); m._backward_impl(st, grad)
        torch.cuda.synchronize()
        print('after', i + 1, 'batches: size', int(m.size()), 'ms/step on the last 40', round((time.perf_counter() - t0) / 40 * 1e3, 4), flush=True)
PY
