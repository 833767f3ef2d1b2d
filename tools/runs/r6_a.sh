set -x
mkdir -p gpurun_out/a
python -m pytest tests/test_sharded_gpu.py -m gpu -x -q > gpurun_out/a/pytest_sharded.txt 2>&1; echo "rc=$?" >> gpurun_out/a/pytest_sharded.txt
tail -15 gpurun_out/a/pytest_sharded.txt
python bench.py --no-cpu-baseline > gpurun_out/a/bench.json 2> gpurun_out/a/bench.err; tail -c 600 gpurun_out/a/bench.err
python bench.py --force-sharded --no-hstu --no-cpu-baseline > gpurun_out/a/bench_sharded.json 2> gpurun_out/a/bench_sharded.err; tail -c 600 gpurun_out/a/bench_sharded.err
cat gpurun_out/a/bench_sharded.json | tail -1 | cut -c1-1500
