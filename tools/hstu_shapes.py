"""Forward / backward time of the d = 256 attention kernels over a set of batch shapes (one process, HIP events):
C3 dense, dense 32 x 4096, and jagged Zipf(1.2) batches clipped to 4096 / 512 with several seeds -- the shapes the block
block-to-sequence maps were judged on.   python tools/hstu_shapes.py [--heads 4] [--dim 256]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from hstu import hstu_varlen_bwd, hstu_varlen_fwd

ap = argparse.ArgumentParser()
ap.add_argument("--heads", type=int, default=4)
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--seeds", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda")
H, d = a.heads, a.dim


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(name, lengths):
    lengths = np.asarray(lengths, np.int64)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int32, device=dev)
    T, L = int(cu[-1]), int(lengths.max())
    g = torch.Generator(device=dev); g.manual_seed(11)
    q, k, v, do = (torch.empty(T, H, d, device=dev).uniform_(-1, 1, generator=g).bfloat16() for _ in range(4))
    alpha = 1.0 / d ** 0.5
    reps = 20 if T * L < 3e7 else 6
    tf = timeit(lambda: hstu_varlen_fwd(q, k, v, cu, L, L, None, None, 1, True, alpha), reps)
    tb = timeit(lambda: hstu_varlen_bwd(do, q, k, v, cu, L, L, None, None, 1, True, alpha), reps)
    fl = bench.hstu_flops([int(x) for x in lengths], H, d)
    print(f"{name:34s} tokens {T:7d} max {L:5d}  fwd {tf * 1e3:8.1f} us {fl / tf / 1e9:6.0f} TF   bwd {tb * 1e3:8.1f} us {2.5 * fl / tb / 1e9:6.0f} TF", flush=True)


run("C3 dense 32 x 512", [512] * 32)
run("dense 32 x 4096", [4096] * 32)
run("dense 8 x 4096", [4096] * 8)
for seed in range(1, a.seeds + 1):
    rng = np.random.default_rng(seed)
    run(f"jagged zipf(1.2) <= 4096, seed {seed}", np.clip(rng.zipf(1.2, 32) + 31, 32, 4096))
rng = np.random.default_rng(1)
run("jagged zipf(1.2) <= 512", np.clip(rng.zipf(1.2, 32) + 31, 32, 512))
rng = np.random.default_rng(7)
run("jagged uniform 64..2048 x 64", rng.integers(64, 2049, 64))
