mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/b10.json 2> gpurun_out/b10.err
tail -3 gpurun_out/b10.err; python -c "
import json; d=json.loads(open('gpurun_out/b10.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','sustained','step_roofline')}); print(d['roofline']); print(d['cpu_baseline']); print(d['hstu']['fwd_ms'], d['hstu']['bwd_ms'])"
