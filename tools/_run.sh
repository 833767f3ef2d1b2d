R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_hstu_gpu.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['hstu']; print(h['value'], h['fwd_ms'], h['bwd_ms'], h['fwd_TFLOPs'], h['bwd_TFLOPs'])"
timeout 300 python tools/bench_extended.py --only hstu_dense_32x4096,hstu_jagged_zipf_32_4096,hstu_jagged_zipf_32_512 2>&1 | tail -1
