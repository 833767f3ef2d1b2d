"""A/B of MI355_FUSED_FASTMOD (the probe kernel's bucket arithmetic without 64-bit divisions) on the C2 workload, in one
process: (1) a table filled through the fast path must be found, key for key, by the generic `table_lookup` (same buckets);
(2) ms per fwd+bwd step with the switch off / on, alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch
import bench
import dynamicemb_extensions as e
from dynamicemb.scored_hashtable import ScoreArg

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
NB = 8
batches = bench.zipf_batches(10_000_000, 0.99, 65536, NB, dev, seed=1234)
grad = (torch.randn(65536, 128, device=dev) * 0.01).to(torch.bfloat16)
os.environ["MI355_ENV_LIVE"] = "1"
os.environ["MI355_FUSED_FASTMOD"] = "1"
m = bench.build_module(10_000_000, 128, dev)
m.train()
with torch.no_grad():
    for k, o in batches:
        m._forward_impl(k, o, train=True)
torch.cuda.synchronize()
allk = torch.unique(torch.cat([k for k, _ in batches]))
_, found, _ = m.table.lookup(allk, torch.zeros_like(allk), ScoreArg("score", None, e.ScorePolicy.CONST))
print(f"keys inserted through the fast path: {allk.numel()}  found by the generic lookup: {int(found.sum())}  table size {int(m.table.size())}")

def run(flag, reps=60):
    os.environ["MI355_FUSED_FASTMOD"] = flag
    for i in range(8):
        out, st = m._forward_impl(*batches[i % NB], train=True); m._backward_impl(st, grad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        out, st = m._forward_impl(*batches[i % NB], train=True); m._backward_impl(st, grad)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for flag in ("0", "1", "0", "1"):
    print(f"FASTMOD={flag}: {run(flag) * 1e3:.1f} us per step")
