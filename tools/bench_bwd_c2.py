"""Backward stages of the C2 pattern (Zipf or uniform keys): times `backward_fused` (bwd_kernel) with HIP events and
checks the updated rows against a torch index_add reference.  Usage: bench_bwd_c2.py [iters] [zipf|uniform]
(MI355_LIB selects a library build; NOHOT=1 disables the hot-row list)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch
import dynamicemb_extensions as ext
dev = torch.device("cuda"); rows, D, B = 10_000_000, 128, 65536
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "zipf"
table = torch.empty(rows, D, device=dev).uniform_(-1, 1)
g = torch.Generator(device=dev); g.manual_seed(0)
lens = torch.randint(1, 11, (B,), device=dev, generator=g)
off = torch.zeros(B + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(lens, 0); nt = int(off[-1])
if mode == "zipf":
    w = torch.arange(1, rows + 1, device=dev, dtype=torch.float64).pow_(-0.99); cdf = torch.cumsum(w, 0); cdf /= cdf[-1].clone()
    perm = torch.randperm(rows, device=dev, generator=g)
    keys = perm[torch.searchsorted(cdf, torch.rand(nt, device=dev, dtype=torch.float64, generator=g)).clamp_(max=rows - 1)]
else:
    keys = torch.randperm(rows, device=dev, generator=g)[:nt]
uk, rev = torch.unique(keys, return_inverse=True); rev = rev.contiguous()
addr = table.data_ptr() + uk * (D * 4)
grad = (torch.randn(B, D, device=dev) * 0.01).to(torch.bfloat16)
nu = uk.numel()
use_hot = os.environ.get("NOHOT") is None
# reference of ONE step: rows[uk] -= lr * bf16(sum of the bags' grads)
bag = torch.repeat_interleave(torch.arange(B, device=dev), lens)
ref = torch.zeros(nu, D, device=dev).index_add_(0, rev, grad.float()[bag])
want = table[uk] - 0.1 * ref.bfloat16().float()
times = []
for i in range(iters):
    if use_hot:
        p, c, hot = ext.group_by_unique(rev, nu, off, dim=D)
    else:
        p, c = ext.group_by_unique(rev, nu, off); hot = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ext.backward_fused(p, c, nt, nu, grad, B, D, 0, off, None, addr, torch.float32, 1, lr=0.1, hot=hot)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) * 1e3)
    if i == 0:
        got = table[uk]
        err = (got - want).abs().max().item()
        bad = ((got - want).abs() > 2e-3).any(1).sum().item()   # bf16 rounding of sums that land on a tie may differ
times.sort()
print(f"{os.environ.get('MI355_LIB', 'default')[-16:]:>16s} {mode} nt {nt} nu {nu} bwd_kernel median {times[len(times) // 2]:.1f} us min {times[0]:.1f} us  "
      f"max_err {err:.2e} bad_rows {bad}")
