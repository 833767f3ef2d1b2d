"""Forward-only A/B of the d = 256 attention kernels over a few shapes, one process per library / environment variant
(MI355_LIB, MI355_HSTU_PC are read once per process).  Prints the time and a checksum of the output bits --
variants of one algorithm (same MFMA order, same roundings) must print the same checksum.
    python tools/hstu_fwd_ab.py [--shapes c3,d4096,d8x4096,jag1,jag2] [--reps N]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from hstu import hstu_varlen_fwd

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="c3,d4096,d8x4096,jag1")
ap.add_argument("--heads", type=int, default=4)
ap.add_argument("--reps", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda")
H, d = a.heads, 256


def shape(name):
    if name == "c3": return [512] * 32
    if name == "d4096": return [4096] * 32
    if name.startswith("d") and "x" in name:
        n, l = name[1:].split("x"); return [int(l)] * int(n)
    if name == "d1024": return [1024] * 32
    if name.startswith("jag"):
        rng = np.random.default_rng(int(name[3:]))
        return list(np.clip(rng.zipf(1.2, 32) + 31, 32, 4096))
    if name == "ragged": return [1, 63, 64, 65, 127, 128, 129, 500, 1000, 31, 257, 4095]
    raise SystemExit(name)


tag = f"lib={os.path.basename(os.environ.get('MI355_LIB', 'default'))} PC={os.environ.get('MI355_HSTU_PC', '-')}"
for name in a.shapes.split(","):
    lengths = np.asarray(shape(name), np.int64)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int32, device=dev)
    T, L = int(cu[-1]), int(lengths.max())
    g = torch.Generator(device=dev); g.manual_seed(11)
    q, k, v = (torch.empty(T, H, d, device=dev).uniform_(-1, 1, generator=g).bfloat16() for _ in range(3))
    alpha = 1.0 / d ** 0.5
    reps = a.reps or (30 if T * L < 3e7 else 8)
    fn = lambda: hstu_varlen_fwd(q, k, v, cu, L, L, None, None, 1, True, alpha)
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    bits = out.view(torch.int16).to(torch.int64)
    w = (torch.arange(bits.numel(), device=dev, dtype=torch.int64) % 8191 + 1).view_as(bits)
    chk = int((bits * w).sum().item()) & 0xffffffffffff
    fl = bench.hstu_flops([int(x) for x in lengths], H, d)
    nan = bool(torch.isnan(out.float()).any())
    print(f"{tag:48s} {name:8s} fwd {t * 1e3:8.1f} us {fl / t / 1e9:6.0f} TF  chk {chk:012x}{'  NaN!' if nan else ''}", flush=True)
