R=$GRAFT_REPO_ROOT
cd $R
run() { env "$@" timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hstu 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['sustained']['ms_per_step']*1000,1), round(d['roofline']['kernels']['bwd_kernel']['ms']*1000,1))"; }
run A=0
run MI355_HOT=3
run MI355_HOT=6
run MI355_HOT=8
run MI355_WAVE=32
run MI355_WAVE=128
run MI355_CHUNK=512
run MI355_CHUNK=2048
run MI355_HOT_BLOCKS=1024
run MI355_WAVE_BLOCKS=2048
run A=0
