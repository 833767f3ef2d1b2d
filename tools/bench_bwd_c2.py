"""Runs the backward stages of the C2 (Zipf) pattern a few times (for rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch
import dynamicemb_extensions as ext
dev = torch.device("cuda"); rows, D, B = 10_000_000, 128, 65536
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
print("nt", nt, "nu", nu, mode)
use_hot = os.environ.get("NOHOT") is None
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    if use_hot:
        p, c, hot = ext.group_by_unique(rev, nu, off, dim=D)
    else:
        p, c = ext.group_by_unique(rev, nu, off); hot = None
    ext.backward_fused(p, c, nt, nu, grad, B, D, 0, off, None, addr, torch.float32, 1, lr=0.1, hot=hot)
torch.cuda.synchronize()
