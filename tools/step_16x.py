"""The C2 step at 16x the batch (B = 1,048,576 bags, ~5.8 M keys) on the benchmark's table, a few steps in the steady state -- the
command the 16x kernel trace / PMC passes of a round profile (tools/runs/r5_*.sh).  Prints ms per step.
    python tools/step_16x.py [--steps 6] [--mult 16]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--mult", type=int, default=16)
ap.add_argument("--rows", type=int, default=10_000_000)
a = ap.parse_args()
dev = torch.device("cuda")
B = a.mult * 65536
batches = bench.zipf_batches(a.rows, 0.99, B, 3, dev, seed=777)
module = bench.build_module(a.rows, 128, dev)
module.train()
grad = (torch.randn(B, 128, device=dev) * 0.01).to(torch.bfloat16)
with torch.no_grad():
    for k, o in batches:
        module._forward_impl(k, o, train=True)
for k, o in batches:
    out, st = module._forward_impl(k, o, train=True)
    module._backward_impl(st, grad)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    k, o = batches[i % len(batches)]
    out, st = module._forward_impl(k, o, train=True)
    module._backward_impl(st, grad)
torch.cuda.synchronize()
print(f"16x step: {(time.perf_counter() - t0) / a.steps * 1e3:.4f} ms  ({batches[0][0].numel()} keys, {int(st.uoff[-1])} unique rows)")
