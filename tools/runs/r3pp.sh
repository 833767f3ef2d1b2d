#!/bin/bash
mkdir -p gpurun_out/r3pp
timeout 300 python tools/pipeline_step.py > gpurun_out/r3pp/out.txt 2>&1
grep -v amdgpu gpurun_out/r3pp/out.txt | tail -8
