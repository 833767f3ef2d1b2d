mkdir -p gpurun_out/g
for v in "" _stub _noinl "" _stub _noinl; do
  L=$PWD/recsys-examples_amd/lib/librecsys_amd$v.so
  MI355_LIB=$L python bench.py --no-cpu-baseline --no-hstu --no-extra --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant %-8s step %.4f sustained %.4f  gather %.1f us  bwd %.1f us' % ('$v' or 'default', d['ms_per_step'], d['sustained']['ms_per_step'], 1e3*d['roofline']['kernels']['gather_pooled_late_kernel']['ms'], 1e3*d['roofline']['kernels']['bwd_kernel']['ms']))"
done | tee gpurun_out/g/late_variants.txt
