R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_hstu_gpu.py -q -k "exchange" 2>&1 | grep -E "passed|failed|FAILED|^E" | tail -12
