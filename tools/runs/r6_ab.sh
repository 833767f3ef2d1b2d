for i in 1 2 3; do for v in "" _nokl _kl2; do
  MI355_LIB=$PWD/recsys-examples_amd/lib/librecsys_amd$v.so python bench.py --no-cpu-baseline --no-hstu --no-extra --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant %-8s step %.4f sustained %.4f ' % ('$v' or 'keyline1', d['ms_per_step'], d['sustained']['ms_per_step']), {k: round(1e3*v['ms'],1) for k,v in d['roofline']['kernels'].items()})"
done; done
MI355_LIB=$PWD/recsys-examples_amd/lib/librecsys_amd_stampskl2.so timeout 300 python tools/index_phase_stamps.py 2>&1 | grep -v amdgpu.ids | head -15
MI355_LIB=$PWD/recsys-examples_amd/lib/librecsys_amd_kl2.so python -m pytest tests/test_fused_fwd_gpu.py -m gpu -x -q 2>&1 | tail -2
