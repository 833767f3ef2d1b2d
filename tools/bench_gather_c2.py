"""C2 (Zipf) pattern of the fused gather+pool kernel: event-timed, checked against torch (the library's default pooled gather
variant, see value_ops.hip).  Usage: bench_gather_c2.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch
import dynamicemb_extensions as ext
dev = torch.device("cuda"); rows, D, B = 10_000_000, 128, 65536
table = torch.empty(rows, D, device=dev).uniform_(-1, 1)
g = torch.Generator(device=dev); g.manual_seed(0)
lens = torch.randint(1, 11, (B,), device=dev, generator=g)
off = torch.zeros(B + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(lens, 0); nt = int(off[-1])
w = torch.arange(1, rows + 1, device=dev, dtype=torch.float64).pow_(-0.99); cdf = torch.cumsum(w, 0); cdf /= cdf[-1].clone()
perm = torch.randperm(rows, device=dev, generator=g)
keys = perm[torch.searchsorted(cdf, torch.rand(nt, device=dev, dtype=torch.float64, generator=g)).clamp_(max=rows - 1)]
uk, rev = torch.unique(keys, return_inverse=True); rev = rev.contiguous()
addr = table.data_ptr() + uk * (D * 4)
out = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
bag = torch.repeat_interleave(torch.arange(B, device=dev), lens)
want = torch.zeros(B, D, device=dev).index_add_(0, bag, table[keys])
times = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ext.gather_embedding_pooled(None, out, rev, off, 0, D, B, max_D=D, row_addr=addr, src_dtype=torch.float32)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) * 1e3)
times.sort()
err = (out.float() - want).abs().max().item()
exact = torch.equal(out, want.bfloat16())
print(f"nt {nt} nu {uk.numel()} gather median {times[len(times) // 2]:.1f} us min {times[0]:.1f} us  "
      f"max_err {err:.2e} exact_vs_sequential_fp32 {exact}")
