#!/bin/bash
mkdir -p gpurun_out/r3s3
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29534
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_plugin_surface_gpu.py -x -q -m gpu > gpurun_out/r3s3/tests.txt 2>&1
tail -3 gpurun_out/r3s3/tests.txt
for i in 1 2; do
for v in 0 1; do
  MI355_SCAN3=$v timeout 300 python bench.py --force-sharded --shard-mode partial --steps 200 --warmup 20 --no-hstu --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/r3s3/bench_partial_scan3_$v.json 2> gpurun_out/r3s3/err_$v.txt
  python -c "import json,sys; d=json.loads(open('gpurun_out/r3s3/bench_partial_scan3_$v.json').read().strip().splitlines()[-1]); print('partial scan3=$v', d['ms_per_step'])"
done
done
timeout 300 python bench.py --force-sharded --shard-mode rows --steps 200 --warmup 20 --no-hstu --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/r3s3/bench_rows.json 2> gpurun_out/r3s3/err_rows.txt
python -c "import json,sys; d=json.loads(open('gpurun_out/r3s3/bench_rows.json').read().strip().splitlines()[-1]); print('rows', d['ms_per_step'])"
bash tools/prof_sharded.sh > /dev/null 2>&1
mv gpurun_out/sh_* gpurun_out/r3s3/ 2>/dev/null
tail -42 gpurun_out/r3s3/sh_partial_timeline.txt | head -24 | cut -c1-120
