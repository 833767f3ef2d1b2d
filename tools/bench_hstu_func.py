"""Attention under arbitrary mask functions (`func`): the functions read inside the kernels (default) against the dense 0 / -1e9 bias
statement of the same mask (MI355_HSTU_FUNC_DENSE=1; read once per process): time of forward / backward and peak memory.
    python tools/bench_hstu_func.py [--batch 8] [--seqlen 4096]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hstu import hstu_attn_varlen_func

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--seqlen", type=int, default=4096)
ap.add_argument("--heads", type=int, default=4); ap.add_argument("--dim", type=int, default=256); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--mask", default="causal_band", help="causal_band: j <= i plus a band further left (= the causal mask); sink_window: the first 64 keys + "
                "a causal window of 256 (a gap of whole tiles between the two for most rows)")
a = ap.parse_args()
dev = torch.device("cuda")
B, L, H, d = a.batch, a.seqlen, a.heads, a.dim
T = B * L
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = (torch.randn(T, H, d, device=dev, generator=g).mul_(0.5).to(torch.bfloat16).requires_grad_(True) for _ in range(3))
dout = torch.randn(T, H, d, device=dev, generator=g).to(torch.bfloat16)
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=dev)
pos = torch.arange(T, device=dev) % L
f = torch.zeros(1, 3, T, dtype=torch.int32, device=dev)       # a causal prefix + one band further left: j <= i, or i - 1536 <= j < i - 1024
if a.mask == "sink":
    f[0, 0] = torch.minimum(pos + 1, torch.full_like(pos, 64)).to(torch.int32)
elif a.mask == "window_band":
    f[0, 1] = (pos - 255).clamp(min=0).to(torch.int32); f[0, 2] = (pos + 1).to(torch.int32)
elif a.mask == "sink_window":
    f[0, 0] = torch.minimum(pos + 1, torch.full_like(pos, 64)).to(torch.int32); f[0, 1] = (pos - 255).clamp(min=0).to(torch.int32); f[0, 2] = (pos + 1).to(torch.int32)
else:
    f[0, 0] = (pos + 1).to(torch.int32); f[0, 1] = (pos - 1536).clamp(min=0).to(torch.int32); f[0, 2] = (pos - 1024).clamp(min=0).to(torch.int32)
def step():
    out = hstu_attn_varlen_func(q, k, v, cu, cu, None, None, L, L, L, None, None, window_size=(-1, -1), alpha=1.0 / d ** 0.5, func=f)
    out.backward(dout)
    return out
torch.cuda.synchronize(); base = torch.cuda.memory_allocated(); torch.cuda.reset_peak_memory_stats()
out = step(); torch.cuda.synchronize()
peak = torch.cuda.max_memory_allocated() - base
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(a.reps):
    q.grad = k.grad = v.grad = None
    e[0].record()
    out = hstu_attn_varlen_func(q, k, v, cu, cu, None, None, L, L, L, None, None, window_size=(-1, -1), alpha=1.0 / d ** 0.5, func=f)
    e[1].record()
    out.backward(dout)
    e[2].record(); torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
mode = "dense bias" if os.environ.get("MI355_HSTU_FUNC_DENSE") == "1" else "in the kernels"
print(f"func {mode} [{a.mask}]: batch {B} x L {L}, H {H}, d {d}: forward {tf / a.reps:.3f} ms, backward {tb / a.reps:.3f} ms, peak extra memory of a step {peak / 2**20:.0f} MiB "
      f"(q + k + v + dout = {4 * T * H * d * 2 / 2**20:.0f} MiB), checksum {float(out.float().abs().sum()):.6e}")
