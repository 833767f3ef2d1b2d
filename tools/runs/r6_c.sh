mkdir -p gpurun_out/c
python tools/pipeline_step.py > gpurun_out/c/pipe.txt 2>&1
MI355_PREFETCH_C=0 python tools/pipeline_step.py > gpurun_out/c/pipe_pin.txt 2>&1
cat gpurun_out/c/pipe.txt gpurun_out/c/pipe_pin.txt | grep -v amdgpu.ids
