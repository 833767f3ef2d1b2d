"""HSTU attention kernel benchmark (C3 shape by default: B=32, L=512, H=4, d=256, bf16, causal)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hstu import hstu_varlen_fwd, hstu_varlen_bwd
from oracle.hstu_oracle import attn_flops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--seqlen", type=int, default=512)
ap.add_argument("--heads", type=int, default=4); ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--jagged", action="store_true"); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--ctx-targets", type=int, nargs=3, default=None, metavar=("CTX", "TGT", "GROUP"),
                help="contextual rows / target rows per sequence and the target group size (the ranking model's mask)")
ap.add_argument("--window", type=int, nargs=2, default=None, help="local window (left right): times the windowed kernels too")
a = ap.parse_args()
dev = torch.device("cuda")
rng = np.random.default_rng(0)
lengths = np.full(a.batch, a.seqlen) if not a.jagged else np.clip((rng.zipf(1.2, a.batch) * 32), 32, a.seqlen)
off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
T = int(off[-1])
cu = torch.from_numpy(off.astype(np.int32)).to(dev)
q, k, v, do = (torch.empty(T, a.heads, a.dim, device=dev).uniform_(-1, 1).bfloat16() for _ in range(4))
alpha = 1.0 / a.dim ** 0.5
fl = attn_flops(off, a.heads, a.dim, True)
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
nc = nt = None
grp = 1
if a.ctx_targets is not None:
    nc = torch.full((a.batch,), a.ctx_targets[0], dtype=torch.int32, device=dev)
    nt = torch.full((a.batch,), a.ctx_targets[1], dtype=torch.int32, device=dev)
    grp = a.ctx_targets[2]
tf = timeit(lambda: hstu_varlen_fwd(q, k, v, cu, a.seqlen, a.seqlen, nc, nt, grp, True, alpha))
tb = timeit(lambda: hstu_varlen_bwd(do, q, k, v, cu, a.seqlen, a.seqlen, nc, nt, grp, True, alpha))
print(f"T={T} H={a.heads} d={a.dim} causal{' ctx/targets ' + str(a.ctx_targets) if a.ctx_targets else ''}  fwd {tf*1e3:.1f} us  {fl/tf/1e9:.1f} TFLOP/s  {T/tf/1e3:.3e} tok/s | "
      f"bwd {tb*1e3:.1f} us  {2.5*fl/tb/1e9:.1f} TFLOP/s | fwd+bwd {T/(tf+tb)/1e3:.3e} tok/s")
if a.window is not None:
    from hstu.hstu_attn_interface import HstuAttnWindowFunc
    wl, wr = a.window
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    def fwd_bwd():
        out = HstuAttnWindowFunc.apply(qq, kk, vv, cu, a.seqlen, a.seqlen, wl, wr, alpha)
        out.backward(do)
    def fwd_only():
        with torch.no_grad():
            HstuAttnWindowFunc.apply(q, k, v, cu, a.seqlen, a.seqlen, wl, wr, alpha)
    tw, twf = timeit(fwd_bwd), timeit(fwd_only)
    print(f"window ({wl}, {wr}): fwd {twf*1e3:.1f} us  fwd+bwd {tw*1e3:.1f} us (autograd included)  vs causal fwd {tf*1e3:.1f} + bwd {tb*1e3:.1f} us")
