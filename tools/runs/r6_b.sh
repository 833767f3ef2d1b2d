set -x
mkdir -p gpurun_out/b
python -m pytest tests/test_fused_fwd_gpu.py -m gpu -x -q > gpurun_out/b/pytest_fused.txt 2>&1; echo "rc=$?" >> gpurun_out/b/pytest_fused.txt
tail -30 gpurun_out/b/pytest_fused.txt
python bench.py --no-cpu-baseline --no-hstu --no-extra > gpurun_out/b/bench.json 2> gpurun_out/b/bench.err; tail -c 600 gpurun_out/b/bench.err
cut -c1-700 gpurun_out/b/bench.json
