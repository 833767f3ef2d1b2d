"""Phase break-down of the C2 step's kernels from the s_memtime stamps of a profiling build (csrc/stamps.h).

    python recsys-examples_amd/build.py --variant stamps -DMI355_STAMPS=1
    MI355_LIB=$PWD/recsys-examples_amd/lib/librecsys_amd_stamps.so python tools/index_phase_stamps.py [--batch 65536]

Per kernel: average shader cycles between consecutive stamps of thread 0 of every block (the stamp BEFORE a barrier and the one
AFTER it separate a phase's own work from the wait for the block's slowest wave), block life times, and when blocks start and end
on the constant 100 MHz wall clock (first start = 0)."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import mi355_native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--batches", type=int, default=6, help="distinct batches run before the stamped step (300: the table is 3/4 full, "
                "new keys meet full buckets: the eviction regime)")
a = ap.parse_args()
dev = torch.device("cuda")
batches = bench.zipf_batches(a.rows, 0.99, a.batch, a.batches, dev)
module = bench.build_module(a.rows, 128, dev)
module.train()
grad = (torch.randn(a.batch, 128, device=dev) * 0.01).to(torch.bfloat16)
if a.batches <= 6:
    with torch.no_grad():
        for k, o in batches:
            module._forward_impl(k, o, train=True)
for k, o in batches:
    out, st = module._forward_impl(k, o, train=True)
    module._backward_impl(st, grad)
torch.cuda.synchronize()
lib = ctypes.CDLL(mi355_native.LIB_PATH)

PROBE = ["start -> LDS init + key loads issued (sync)", "hash + mod + digest prefetch issue", "wait: barrier", "LDS dedup (CAS / count)",
         "wait: barrier", "partition hist + global atomic + first key-word load issue", "wait: barrier", "probe (key compare, scores, s_tab)",
         "wait: barrier", "records out", "per-occurrence outputs + init rows"]
PART = ["start -> record loads issued + LDS init", "wait: barrier", "merge (LDS hash insert + counts)", "wait: barrier",
        "deferred / hot counts / publish sums", "local scans", "look-back", "outputs", "-"]
SCAT = ["start -> index hops issued + bag range search", "wait: barrier", "bag marks + max-scan", "wait for index hops", "stores"]


SPANS = []


def dump(fn, nblk, nph, names, title, nlive=None):
    f = getattr(lib, fn, None)
    if f is None:
        print(f"{title}: no {fn} in this library (not a stamps build)")
        return
    buf = np.zeros((nblk, nph + 2), np.uint64)
    f.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    live = buf[:, 0] != 0
    d = buf[live].astype(np.float64)
    if d.shape[0] == 0:
        print(f"{title}: no stamps recorded")
        return
    print(f"== {title}: {d.shape[0]} blocks stamped")
    life = d[:, nph - 1] - d[:, 0]
    if names:
        for i, nm in enumerate(names):
            if i + 1 >= nph:
                break
            x = d[:, i + 1] - d[:, i]
            print(f"   {nm:62s} avg {x.mean():8.0f}  p50 {np.median(x):8.0f}  max {x.max():8.0f} cyc  ({100 * x.sum() / life.sum():5.1f} %)")
    print(f"   block life (shader cycles): avg {life.mean():.0f}  p50 {np.median(life):.0f}  max {life.max():.0f}")
    w0, w1 = d[:, nph], d[:, nph + 1]
    ok = w1 > 0
    recent = w0 > w0.max() - 6000        # (blocks only an earlier launch used keep their old stamps: drop what is > 60 us older)
    d, w0, w1, ok, life = d[recent], w0[recent], w1[recent], ok[recent], life[recent]
    t0 = w0.min()
    SPANS.append((title.split(":")[0].split(" (")[0], w0.min() / 100.0, w1[ok].max() / 100.0))
    print(f"   wall clock (us, 100 MHz): starts {((w0 - t0) / 100).min():.2f} .. {((w0 - t0) / 100).max():.2f}  "
          f"ends p10 {np.percentile((w1[ok] - t0) / 100, 10):.2f}  p50 {np.median((w1[ok] - t0) / 100):.2f}  p90 {np.percentile((w1[ok] - t0) / 100, 90):.2f}  "
          f"max {((w1[ok] - t0) / 100).max():.2f};  block wall life avg {((w1[ok] - w0[ok]) / 100).mean():.2f} us"
          f"  -> {life[ok].mean() / ((w1[ok] - w0[ok]) / 100).mean() / 1e3:.2f} GHz")
    if not names:
        st = np.sort((w0 - t0) / 100)
        en = np.sort((w1[ok] - t0) / 100)
        qs = [0, 10, 25, 50, 75, 90, 100]
        print("   start percentiles (us):", " ".join(f"{np.percentile(st, q):.1f}" for q in qs))
        print("   end   percentiles (us):", " ".join(f"{np.percentile(en, q):.1f}" for q in qs))


PROBE_C = ['start -> loads issued + LDS init + barrier 0', 'hash + bucket + digest prefetch issue + bag marks', 'wait: barrier A', 'LDS dedup (+ pair histogram) + bag scan half 1', 'wait: barrier B', 'reservation issued, reps, list starts, 2 key words issued, bag scan half 2', 'wait: barrier C', 'list starts, probe resolve, scores, reserved bases', 'wait: barrier D', '-', 'records + per-occurrence outputs']
if True:
    dump("mi355_debug_stamps_probe", 1024, 12, PROBE_C, "probe_c_kernel")
else:
    dump("mi355_debug_stamps_probe", 1024, 12, PROBE, "fused_probe_kernel")
if os.environ.get("MI355_FUSED_PART", "2") == "1":
    dump("mi355_debug_stamps_part", 1024, 10, PART, "fused_part_kernel")
    dump("mi355_debug_stamps_scatter", 2048, 6, SCAT, "csr_scatter_kernel")
PART2 = ["start -> record loads issued + LDS init", "wait: barrier", "merge (LDS hash insert + counts + rank bases out)", "wait: barrier",
         "entry scan + publish sums", "look-back", "outputs per unique row", "outputs per record (CSR entries)", "-"]
PART_L = ["init + barrier", "merge pass", "wait: barrier", "eviction check, entry sums, block scan, publish", "look-back",
          "unique-row outputs (+ keys)", "wait: barrier", "output pass (CSR entries)", "tail", "-"]
EVICT = ["entry -> bucket lock taken", "re-probe of the bucket", "score scan: the lane's minimum + its eligibility",
         "group arg-min, lock / digest / score stores", "row initialisation issued", "(drain) + key published", "LDS hash insert, record, unlock",
         "rest of the pass (other keys)", "-"]
if os.environ.get("MI355_FUSED_PART", "2") != "1":
    dump("mi355_debug_stamps_evict", 1024, 10, EVICT, "part_evict: FIRST deferred key of every partition block that had one (any step so far)")
if False:
    dump("mi355_debug_stamps_part", 1024, 10, PART_L[:9], "part3_lean (partition role of the gather's launch)")
elif os.environ.get("MI355_FUSED_PART", "2") != "1":
    dump("mi355_debug_stamps_part", 1024, 10, PART2, "fused_part3_kernel")
    dump("mi355_debug_stamps_fgather", 16384, 2, None, "gather_pooled_late_kernel (thread 0 of each block)")
dump("mi355_debug_stamps_gather", 16384, 2, None, "gather_pooled_pipe_kernel (thread 0 of each block)")
dump("mi355_debug_stamps_bwd", 16384, 2, None, "bwd_kernel (thread 0 of each block)")

print("== absolute wall clock of the last step (us): first block start .. last block end, and the gap to the previous kernel")
SPANS.sort(key=lambda x: x[1])
base = SPANS[0][1] if SPANS else 0
prev_end = None
for name, a0, a1 in SPANS:
    gap = "" if prev_end is None else f"   gap after previous: {a0 - prev_end:6.2f}"
    print(f"   {name:60s} {a0 - base:8.2f} .. {a1 - base:8.2f}{gap}")
    prev_end = a1
