"""Phase stamps of the big-batch stage's kernels at the 16x batch (a -DMI355_STAMPS=1 build, MI355_LIB=...): probe_c_kernel<kStage>
and fused_part3s_kernel (thread 0 of the first 1 024 blocks)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, mi355_native
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
B = mult * 65536
batches = bench.zipf_batches(10_000_000, 0.99, B, 2, dev, seed=777)
module = bench.build_module(10_000_000, 128, dev); module.train()
grad = (torch.randn(B, 128, device=dev) * 0.01).to(torch.bfloat16)
with torch.no_grad():
    for k, o in batches: module._forward_impl(k, o, train=True)
for _ in range(2):
    for k, o in batches:
        out, st = module._forward_impl(k, o, train=True); module._backward_impl(st, grad)
torch.cuda.synchronize()
lib = ctypes.CDLL(mi355_native.LIB_PATH)
def dump(fn, nblk, nph, title, names):
    f = getattr(lib, fn); buf = np.zeros((nblk, nph + 2), np.uint64)
    f.argtypes = [ctypes.c_void_p, ctypes.c_longlong]; assert f(buf.ctypes.data, buf.nbytes) == 0
    d = buf[buf[:, 0] != 0].astype(np.float64)
    print(f"== {title}: {d.shape[0]} blocks stamped")
    life = d[:, nph - 1] - d[:, 0]
    for i, nm in enumerate(names):
        x = d[:, i + 1] - d[:, i]
        print(f"   {nm:50s} avg {x.mean():9.0f} p50 {np.median(x):9.0f} max {x.max():9.0f} cyc ({100 * x.sum() / life.sum():5.1f} %)")
    w0, w1 = d[:, nph], d[:, nph + 1]; t0 = w0.min()
    print(f"   block life avg {life.mean():.0f} max {life.max():.0f} cyc; wall: starts {((w0 - t0) / 100).min():.1f}..{((w0 - t0) / 100).max():.1f} us, ends p50 {np.median((w1 - t0) / 100):.1f} max {((w1 - t0) / 100).max():.1f} us; block wall life avg {((w1 - w0) / 100).mean():.1f} max {((w1 - w0) / 100).max():.1f} us")
dump("mi355_debug_stamps_probe", 1024, 12, "probe_c_kernel<kStage> (first 1024 tiles)", ["load issue + LDS init + barrier 0", "hash + digest issue + marks", "barrier A", "dedup + bag scan 1", "barrier B", "phase 3", "barrier C", "phase 4 (resolve, scores)", "barrier D", "-", "outputs"])
dump("mi355_debug_stamps_part", 1024, 10, "fused_part3s_kernel", ["init + barrier", "merge pass", "barrier", "eviction + entry scan + publish", "look-back", "unique-row outputs", "output pass", "tail", "-"])
