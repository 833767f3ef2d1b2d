"""Microbenchmark of the fused gather+pool kernel on different access patterns (MI355X)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import dynamicemb_extensions as ext

dev = torch.device("cuda")
rows, D, B = 10_000_000, 128, 65536
table = torch.empty(rows, D, device=dev).uniform_(-1, 1)
g = torch.Generator(device=dev); g.manual_seed(0)

def run(name, rev, offsets, addr, reps=20):
    out = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ext.gather_embedding_pooled(None, out, rev, offsets, 0, D, B, max_D=D, row_addr=addr, src_dtype=torch.float32)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ext.gather_embedding_pooled(None, out, rev, offsets, 0, D, B, max_D=D, row_addr=addr, src_dtype=torch.float32)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    nt = rev.numel(); nu = addr.numel()
    ms = float(np.median(ts))
    print(f"{name:40s} nt={nt} nu={nu} {ms*1e3:8.1f} us  rows-read {nt*512/ms/1e6:8.0f} GB/s  unique {nu*512/ms/1e6:8.0f} GB/s")

def offsets_for(lens):
    o = torch.zeros(B + 1, dtype=torch.int64, device=dev); o[1:] = torch.cumsum(lens, 0); return o

for L in (1, 4, 8):
    lens = torch.full((B,), L, dtype=torch.int64, device=dev)
    off = offsets_for(lens); nt = B * L
    # (a) sequential rows, all distinct
    addr = table.data_ptr() + torch.arange(nt, device=dev, dtype=torch.int64) * (D * 4)
    rev = torch.arange(nt, device=dev, dtype=torch.int64)
    run(f"L={L} sequential distinct rows", rev, off, addr)
    # (b) random distinct rows
    slots = torch.randperm(rows, device=dev, generator=g)[:nt]
    addr = table.data_ptr() + slots * (D * 4)
    run(f"L={L} random distinct rows", rev, off, addr)
    # (c) random distinct rows, random reverse index
    rev2 = torch.randperm(nt, device=dev, generator=g)
    run(f"L={L} random rows + random rev", rev2, off, addr)

lens = torch.randint(1, 11, (B,), device=dev, generator=g)
off = offsets_for(lens); nt = int(off[-1])
w = torch.arange(1, rows + 1, device=dev, dtype=torch.float64).pow_(-0.99); cdf = torch.cumsum(w, 0); cdf /= cdf[-1].clone()
perm = torch.randperm(rows, device=dev, generator=g)
keys = perm[torch.searchsorted(cdf, torch.rand(nt, device=dev, dtype=torch.float64, generator=g)).clamp_(max=rows - 1)]
uk, rev = torch.unique(keys, return_inverse=True)
addr = table.data_ptr() + uk * (D * 4)
run("C2 zipf batch (randint(1,11) bags)", rev.contiguous(), off, addr)
