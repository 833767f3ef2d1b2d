#!/bin/bash
# round 4: one-kernel eval for several tables / sequence lookups -- tests, then the eval forward of the model shapes A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4mt; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fused_fwd_gpu.py tests/test_module_gpu.py -q -m gpu -x > $O/pytest_fused.txt 2>&1; tail -15 $O/pytest_fused.txt
timeout 300 python tools/bench_model_shapes.py --steps 100 --eval > $O/shapes_eval.txt 2>&1; grep -v amdgpu.ids $O/shapes_eval.txt | cut -c1-200
MI355_EVAL_FUSED_MT=0 timeout 300 python tools/bench_model_shapes.py --steps 100 --eval > $O/shapes_eval0.txt 2>&1; grep -v amdgpu.ids $O/shapes_eval0.txt | cut -c1-200
timeout 300 python tools/bench_model_shapes.py --steps 50 > $O/shapes_mt.txt 2>&1; grep -v amdgpu.ids $O/shapes_mt.txt | cut -c1-200
