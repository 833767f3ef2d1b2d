#!/bin/bash
mkdir -p gpurun_out/r3soak
export MASTER_ADDR=127.0.0.1
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r3soak/run_$i.txt 2>&1
  tail -1 gpurun_out/r3soak/run_$i.txt
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/r3soak/bench.json 2> gpurun_out/r3soak/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r3soak/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], d['hstu_jagged']['fwd_ms'], d['cpu_baseline']['value'])"
