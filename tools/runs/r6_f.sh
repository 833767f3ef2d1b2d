mkdir -p gpurun_out/f
python -m pytest tests -m gpu -x -q > gpurun_out/f/pytest_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/f/pytest_gpu.txt
tail -25 gpurun_out/f/pytest_gpu.txt
python bench.py --no-cpu-baseline > gpurun_out/f/bench.json 2> gpurun_out/f/bench.err; tail -c 300 gpurun_out/f/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/f/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['sustained']['ms_per_step'], d.get('pipelined_ms_per_step'), {k:v['ms'] for k,v in d['roofline']['kernels'].items()})
print(d['c2_16x']['ms_per_step'], d['hstu']['fwd_ms'], d['hstu']['bwd_ms'], d['hstu_l4096']['fwd_ms'], d['hstu_l4096']['bwd_ms'])
PY
