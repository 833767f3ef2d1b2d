#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo/recsys-examples_amd:/root/repo
timeout 900 python -m pytest tests/test_hstu_gpu.py tests/test_full_size_gpu.py -m gpu -q 2>&1 | tail -8
timeout 200 python tools/bench_hstu.py --reps 50 2>&1 | tail -1
