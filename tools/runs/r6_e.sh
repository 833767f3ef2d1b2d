mkdir -p gpurun_out/e
python -m pytest tests/test_hstu_gpu.py -m gpu -x -q -k "func or mask" > gpurun_out/e/pytest.txt 2>&1; tail -5 gpurun_out/e/pytest.txt
for w in 1 0; do
  MI355_HSTU_WSKIP=$w python tools/bench_hstu_func.py 2>&1 | grep -v amdgpu.ids | sed "s/^/WSKIP=$w /"
  MI355_HSTU_WSKIP=$w python tools/bench_hstu_func.py --batch 32 --seqlen 512 2>&1 | grep -v amdgpu.ids | sed "s/^/WSKIP=$w /"
done | tee gpurun_out/e/func.txt
