#!/bin/bash
mkdir -p gpurun_out/r3sw
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 100 --warmup 20 --no-hstu --no-cpu-baseline --no-extra > gpurun_out/r3sw/$name.json 2> /dev/null
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f'gpurun_out/r3sw/{n}.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print(f"{n:28s} step {d['ms_per_step']*1e3:7.1f} sus {d['sustained']['ms_per_step']*1e3:7.1f}  bwd {k['bwd_kernel']['ms']*1e3:6.1f}  gather {list(k.values())[1]['ms']*1e3:6.1f}")
PY
}
run base A=1
run hot3 MI355_HOT=3
run hot6 MI355_HOT=6
run hot8 MI355_HOT=8
run wave64 MI355_WAVE=64
run wave256 MI355_WAVE=256
run chunk512 MI355_CHUNK=512
run chunk2048 MI355_CHUNK=2048
run hotblk1024 MI355_HOT_BLOCKS=1024
run hotblk4096 MI355_HOT_BLOCKS=4096
run waveblk512 MI355_WAVE_BLOCKS=512
run waveblk2048 MI355_WAVE_BLOCKS=2048
run base2 A=1
