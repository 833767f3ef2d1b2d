mkdir -p gpurun_out/h
for f in 1 1 1; do
  MI355_FUSED=$f python bench.py --no-cpu-baseline --no-hstu --no-extra --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('MI355_FUSED=$f step %.4f sustained %.4f ' % (d['ms_per_step'], d['sustained']['ms_per_step']), {n: round(1e3*v['ms'],1) for n,v in k.items()})"
done | tee gpurun_out/h/late.txt
