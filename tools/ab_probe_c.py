"""A/B of the probe kernels of path (c) inside ONE process on the C2 workload (MI355_PROBE_C is read per call under
MI355_ENV_LIVE=1): 0 = the round-3 kernel, 1 = probe_c_kernel (one block per CU), 2 = its two-keys-per-thread form, 3 = full 1024-key tiles.  Steps are timed in interleaved rounds; the pooled
output of every variant must be bit-identical to variant 0's on the same batch.
    python tools/ab_probe_c.py [--rounds 4] [--variants 0,1,2,3,4]"""
import argparse, os, sys, time
os.environ["MI355_ENV_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--variants", default="0,1,2,3")
ap.add_argument("--var", default="MI355_PROBE_C", help="the per-call knob to switch (any knob the library re-reads under MI355_ENV_LIVE)")
ap.add_argument("--batch", type=int, default=65536)
a = ap.parse_args()
variants = a.variants.split(",")
dev = torch.device("cuda")
batches = bench.zipf_batches(10_000_000, 0.99, a.batch, 40, dev)
module = bench.build_module(10_000_000, 128, dev)
module.train()
grad = (torch.randn(a.batch, 128, device=dev) * 0.01).to(torch.bfloat16)
with torch.no_grad():
    for k, o in batches:
        module._forward_impl(k, o, train=True)
torch.cuda.synchronize()
ref_out = {}
for v in variants:          # identical outputs and unique counts (zero gradient: the rows stay put)
    os.environ[a.var] = v
    for bi in (0, 7):
        k, o = batches[bi]
        out, st = module._forward_impl(k, o, train=True)
        nu = int(st.uoff[-1])
        module._backward_impl(st, torch.zeros_like(grad))
        key = (bi,)
        if key not in ref_out:
            ref_out[key] = (out.clone(), nu)
        else:
            assert torch.equal(ref_out[key][0], out), f"variant {v}: pooled output differs from variant {variants[0]}"
            assert ref_out[key][1] == nu, f"variant {v}: {nu} unique rows vs {ref_out[key][1]}"
print("outputs / unique counts identical over variants", variants)
times = {v: [] for v in variants}
for r in range(a.rounds):
    for v in variants:
        os.environ[a.var] = v
        for k, o in batches[:5]:
            out, st = module._forward_impl(k, o, train=True); module._backward_impl(st, grad)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(3):
            for k, o in batches:
                out, st = module._forward_impl(k, o, train=True); module._backward_impl(st, grad)
        torch.cuda.synchronize()
        times[v].append((time.perf_counter() - t0) / (3 * len(batches)) * 1e3)
for v in variants:
    print(f"{a.var}={v}: ms/step per round " + " ".join(f"{x:.4f}" for x in times[v]) + f"   median {np.median(times[v]):.4f}")
