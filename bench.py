#!/usr/bin/env python
"""bench.py -- DynamicEmb lookup+pool forward/backward on MI355X (BASELINE.json configs[1], "C2").

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch that is already resident in HBM:
forward (dedup -> hash lookup -> [insert + init of unseen keys: none in steady state] -> fused
gather+pool straight from the table rows -> bf16 [B, 128]) and backward (group keys by unique row
-> reduce the bf16 gradients -> SGD in place on the fp32 table rows).

Workload (SURVEY 8(d) "C2"): 1 table x 10,000,000 rows x 128-D fp32, SGD; B = 65,536 bags, bag
length randint(1, 11) (reference harness convention), keys Zipf(0.99) over 10 M ranks mapped through
a fixed permutation (seed 1234); every timed batch is different; all touched keys are pre-inserted
(steady state).  N > 1: row-wise model-parallel shards, one process per GPU, keys all-to-all out /
pooled partial sums back; weak scaling (per-GPU batch fixed).

Prints ONE JSON line (rank 0).  `value` = lookups (keys) per second of the whole job.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "recsys-examples_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--alpha", type=float, default=0.99)
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-hstu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 16x-batch C2 step and the jagged attention shape")
    ap.add_argument("--hstu-batch", type=int, default=32)
    ap.add_argument("--hstu-seqlen", type=int, default=512)
    ap.add_argument("--hstu-heads", type=int, default=4)
    ap.add_argument("--hstu-dim", type=int, default=256)
    ap.add_argument("--hstu-reps", type=int, default=20)
    ap.add_argument("--force-sharded", action="store_true", help="run the row-wise sharded path even at N=1 (debug)")
    ap.add_argument("--shard-mode", default="auto", choices=["auto", "rows", "partial"])
    ap.add_argument("--capacity-factor", type=float, default=0.0,
                    help="sharded path: fixed-capacity key exchange with this factor (0: exact all-to-all-v with one host read)")
    return ap.parse_args()


def zipf_keys(min_val, max_val, exponent, size, device):
    """The reference harness's `zipf()` (corelib/dynamicemb/benchmark/dataset_generator.py:75-103) restated call for call:
    p_r = r^-a over r = 1..n in float64, normalised, cast to float32; the value range shuffled by `randperm`; `size`
    draws with replacement by `multinomial` -- the same three generator calls in the same order, so under the same
    `torch.manual_seed` the key stream is bit-identical to the reference's (pinned by tests/golden/demb_flow_golden.npz,
    which holds what the reference's own function returns on the CPU)."""
    n = max_val - min_val
    probs = 1.0 / torch.arange(1, n + 1, dtype=torch.float64, device=device) ** exponent
    probs = (probs / probs.sum()).float()
    shuffled = torch.arange(min_val, max_val, dtype=torch.long, device=device)[torch.randperm(n, device=device)]
    return shuffled[torch.multinomial(probs, size, replacement=True)]


def zipf_batches(rows, alpha, batch, n_batches, device, seed=1234):
    """Key stream of the reference's dataset_generator.zipf() recipe (benchmark/dataset_generator.py:75-103; `zipf_keys`
    above is its literal restatement): p_r ~ r^-alpha over ranks 1..rows, sampled with replacement, rank -> key by ONE
    fixed permutation for all batches (a stable hot set), drawn by inverse CDF in float64 -- multinomial is
    limited to 2^24 categories and float32 probabilities."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.arange(1, rows + 1, device=device, dtype=torch.float64).pow_(-alpha)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    perm = torch.randperm(rows, device=device, generator=g)
    out = []
    for _ in range(n_batches):
        lens = torch.randint(1, 11, (batch,), device=device, generator=g)
        offsets = torch.zeros(batch + 1, dtype=torch.int64, device=device)
        offsets[1:] = torch.cumsum(lens, 0)
        nt = int(offsets[-1].item())
        u = torch.rand(nt, device=device, dtype=torch.float64, generator=g)
        ranks = torch.searchsorted(cdf, u).clamp_(max=rows - 1)
        out.append((perm[ranks].contiguous(), offsets))
    return out


def build_module(rows, dim, device):
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode,
                                              DynamicEmbPoolingMode, DynamicEmbScoreStrategy, DynamicEmbTableOptions,
                                              EmbOptimType)

    opt = DynamicEmbTableOptions(dim=dim, max_capacity=rows, embedding_dtype=torch.float32, index_type=torch.int64,
                                 score_strategy=DynamicEmbScoreStrategy.TIMESTAMP,
                                 initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM,
                                                                            lower=-0.01, upper=0.01))
    return BatchedDynamicEmbeddingTablesV2([opt], feature_table_map=[0], pooling_mode=DynamicEmbPoolingMode.SUM,
                                           output_dtype=torch.bfloat16, optimizer=EmbOptimType.SGD, learning_rate=0.1,
                                           device=device)


def timed_loop(fn, steps, warm=20):
    fn(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def pipelined_ms(module, batches, grad, steps):
    """The reference's prefetch-pipeline order (PrefetchTrainPipelineSparseDist, examples/commons/pipeline/train_pipeline.py:
    533-692) through the module's PUBLIC calls: prefetch_async(batch k + 1) -- the index stage on the module's prefetch stream,
    ordered behind what the main stream holds, i.e. behind the gather of batch k -- then backward(batch k) on the main stream;
    forward(k + 1) only gathers.  The index stage of the next batch runs under the backward of this one."""
    nb = len(batches)

    def run(n):
        module.prefetch_async(*batches[0])
        for i in range(n):
            k, o = batches[i % nb]
            out = module(k, o)                       # consumes the prefetched state: waits for its mark, gathers
            module.prefetch_async(*batches[(i + 1) % nb])
            out.backward(grad)
        out = module(*batches[n % nb])               # drain the last prefetched state
        out.backward(grad)

    return timed_loop(run, steps)


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_ebc_run(weight, dim, batches_cpu, threads, seconds, lr=0.1):
    """What an unsharded CPU embedding-bag lookup + sparse SGD costs on this host WITHOUT FBGEMM's fused TBE kernels: pooled
    forward = dense `index_add_` of the looked-up rows into their bags, backward = `index_add_` of -lr * (the bag's gradient
    row) into the table rows of its keys (duplicates accumulate inside index_add_; no sparse tensor, no coalesce()).
    -> (keys/s, batches timed, keys timed)"""
    torch.set_num_threads(threads)
    keys_done, spent, iters = 0, 0.0, 0
    while spent < seconds:
        for keys, offsets, bag in batches_cpu:
            B = offsets.numel() - 1
            g = torch.ones(B, dim)
            t = time.perf_counter()
            out = torch.zeros(B, dim).index_add_(0, bag, weight[keys])       # forward: gather + pool
            weight.index_add_(0, keys, g[bag], alpha=-lr)                     # backward: per-key gradient rows, SGD in place
            dt_ = time.perf_counter() - t
            if iters > 0:  # the first iteration is warm-up
                spent += dt_
                keys_done += keys.numel()
            iters += 1
            if spent >= seconds:
                break
    del out
    return (keys_done / spent if spent > 0 else 0.0), iters - 1, keys_done


def _cpu_table(rows, dim):
    return torch.empty(rows, dim).uniform_(-0.01, 0.01)


def _with_bag_ids(batches):
    return [(k, o, torch.repeat_interleave(torch.arange(o.numel() - 1), o[1:] - o[:-1])) for k, o in batches]


def _physical_cores() -> int:
    """Physical cores of this host (SMT siblings counted once): distinct thread_siblings_list entries under sysfs, the
    figure `lscpu` derives its "Core(s) per socket x Socket(s)" from; falls back to os.cpu_count()."""
    import glob

    sibs = set()
    for f in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list"):
        try:
            sibs.add(open(f).read().strip())
        except OSError:
            pass
    n = len(sibs) or (os.cpu_count() or 1)
    try:   # never more threads than this process may run on (cgroup / affinity limits of the box)
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return max(n, 1)


def _cpu_torch_ebc_run(weight, dim, batches_cpu, threads, seconds, lr=0.1):
    """What an unsharded TorchRec CPU EmbeddingBagCollection executes per step: TorchRec's CPU EBC is a dict of
    `torch.nn.EmbeddingBag(mode="sum", include_last_offset=True)` modules; its forward is `F.embedding_bag`, its backward with
    `sparse=True` leaves a sparse gradient of the touched rows and `torch.optim.SGD` applies it row-wise.  Same key stream,
    fwd + bwd + SGD.  -> (keys/s, batches timed, keys timed)"""
    torch.set_num_threads(threads)
    emb = torch.nn.EmbeddingBag(weight.shape[0], dim, mode="sum", sparse=True, include_last_offset=True, _weight=weight)
    opt = torch.optim.SGD(emb.parameters(), lr=lr)
    keys_done, spent, iters = 0, 0.0, 0
    while spent < seconds:
        for keys, offsets, _bag in batches_cpu:
            B = offsets.numel() - 1
            g = torch.ones(B, dim)
            t = time.perf_counter()
            out = emb(keys, offsets)            # F.embedding_bag(mode="sum", include_last_offset=True)
            opt.zero_grad(set_to_none=True)
            out.backward(g)                     # sparse gradient (sparse=True)
            opt.step()                          # sparse SGD
            dt_ = time.perf_counter() - t
            if iters > 0:  # the first iteration is warm-up
                spent += dt_
                keys_done += keys.numel()
            iters += 1
            if spent >= seconds:
                break
    return (keys_done / spent if spent > 0 else 0.0), iters - 1, keys_done


def cpu_baseline(args, batches_cpu):
    """SURVEY 8(d) CPU baseline: the reference's TorchRec CPU EmbeddingBagCollection path on THIS host -- TorchRec itself is
    not installed on the box, so its per-table module is timed directly: `torch.nn.EmbeddingBag(mode="sum",
    include_last_offset=True, sparse=True)` forward + sparse backward + `torch.optim.SGD` (kind "port": the same torch
    operators TorchRec's CPU EBC dispatches to), on the C2 key stream, over a thread sweep {1, 8, 16, 32, 64, PHYSICAL cores};
    `value` is the best leg and `cores` the thread count it used, every leg is in `value_by_threads`.  Second key `index_add_port`: the dense `index_add_`
    restatement of rounds 2-3 at the same thread counts.  About `--cpu-seconds` of host time.  Plus the C1 plumbing
    configuration (BASELINE configs[0]) through the same EmbeddingBag path."""
    ncpu = os.cpu_count() or 1
    phys = _physical_cores()
    t0 = time.time()
    weight = _cpu_table(args.rows, args.dim)
    batches_cpu = _with_bag_ids(batches_cpu)
    # thread sweep {1, 8, 16, 32, 64, physical cores} (round-4 review: the all-cores leg alone is pathological on a 128-core
    # host -- 1.7 M lookups/s against 3.9 M on one thread; the baseline a reader expects is the best of a sweep, all legs shown)
    counts = sorted({t for t in (1, 8, 16, 32, 64, phys) if t <= phys} | {1})
    per, per_port = {}, {}
    sample = []
    for th in counts:
        v, nb, nk = _cpu_torch_ebc_run(weight, args.dim, batches_cpu, th, 0.7 * args.cpu_seconds / len(counts))
        per[th] = v
        sample.append(f"{th} threads: {nb} batches / {nk} keys")
    for th in sorted({1, phys}):
        per_port[th] = _cpu_ebc_run(weight, args.dim, batches_cpu, th, 0.1 * args.cpu_seconds)[0]
    best = max(per, key=per.get)
    # C1 (BASELINE configs[0]): 1 table x 100 K rows x 32-D, batch 512 bags of 1..10 keys, CPU path only
    g = torch.Generator().manual_seed(0)
    c1 = []
    for _ in range(8):
        lens = torch.randint(1, 11, (512,), generator=g)
        off = torch.zeros(513, dtype=torch.int64)
        off[1:] = torch.cumsum(lens, 0)
        c1.append((torch.randint(0, 100_000, (int(off[-1]),), generator=g), off))
    w1 = _cpu_table(100_000, 32)
    c1 = _with_bag_ids(c1)
    c1_per = {th: _cpu_torch_ebc_run(w1, 32, c1, th, 0.05 * args.cpu_seconds)[0] for th in sorted({1, phys})}
    torch.set_num_threads(ncpu)
    return {"value": per[best], "unit": "lookups/s", "cores": best, "kind": "port",
            "what": "torch.nn.EmbeddingBag(mode='sum', include_last_offset=True, sparse=True) fwd + sparse bwd + torch.optim.SGD "
                    "(the operators TorchRec's CPU EmbeddingBagCollection dispatches to; TorchRec is not installed here)",
            "value_by_threads": {str(k): v for k, v in per.items()}, "value_1_thread": per[1], "value_physical_cores": per[phys],
            "physical_cores": phys, "cpu_model": _cpu_model(), "os_cpu_count": ncpu,
            "index_add_port": {"what": "dense index_add_ accumulation per bag / per key (rounds 2-3 baseline)",
                               "value_by_threads": {str(k): v for k, v in per_port.items()}},
            "c1": {"workload": "C1: 1 table x 100000 rows x 32-D fp32, batch 512 bags x randint(1,11) keys, SUM, SGD",
                   "value_by_threads": {str(k): v for k, v in c1_per.items()}, "value": max(c1_per.values()),
                   "unit": "lookups/s"},
            "sample": f"C2 key stream on a {args.rows}x{args.dim} fp32 host table: EmbeddingBag(sum) forward, sparse backward, "
                      f"sparse SGD; " + "; ".join(sample) + f"; {time.time() - t0:.0f} s of host time incl. table setup"}


def kernel_roofline(module, batches, grad, batch, D, e=4, o=2):
    """The two bandwidth kernels of the step, timed live with HIP events on the launch stream inside the REAL step calls:
    the library brackets the gather launch of its fused forward and the fused reduce + optimizer launch of its backward
    (mi355_profile_kernels / mi355_profile_ms).  Each against its own algorithmic bytes
     -> (`roofline` object, step algorithmic bytes)."""
    from mi355_native import check, lib

    L = lib()
    fused = bool(getattr(module, "_fused", False))
    check(L.mi355_profile_kernels(1), "profile_kernels")
    nu_list, fwd_ms, bwd_ms = [], [], []
    for keys, offsets in batches:
        out, st = module._forward_impl(keys, offsets, train=True)
        if fused:
            fwd_ms.append(float(L.mi355_profile_ms(0)))
        nu_list.append((keys.numel(), int(st.uoff[-1].item())))
        module._backward_impl(st, grad)
        bwd_ms.append(float(L.mi355_profile_ms(1)))
    L.mi355_profile_kernels(0)
    nt_avg = float(np.mean([a for a, _ in nu_list]))
    nu_avg = float(np.mean([b for _, b in nu_list]))
    FB = batch
    # gather: row address of every key + offsets + each unique row once (the duplicates are L2 hits) + pooled output
    fwd_bytes = 8 * nt_avg + 8 * (FB + 1) + nu_avg * D * e + FB * D * o
    bwd_bytes = 4 * nt_avg + 4 * (nu_avg + 1) + 8 * nu_avg + FB * D * o + 2 * nu_avg * D * e
    b_ms = float(np.median(bwd_ms))
    kern = {"bwd_kernel": {"ms": b_ms, "algorithmic_bytes": bwd_bytes, "GB/s": bwd_bytes / b_ms / 1e6}}
    if fwd_ms:
        f_ms = float(np.median(fwd_ms))
        kern["gather_pooled_late_kernel"] = {"ms": f_ms, "algorithmic_bytes": fwd_bytes, "GB/s": fwd_bytes / f_ms / 1e6}
    dom = max(kern.items(), key=lambda kv: kv[1]["ms"])
    # HBM traffic of the dominant kernel per launch: PMC counters cannot be read from inside this process; the figure is
    # the one of the committed separate rocprofv3 --pmc passes over THIS command (tools/pmc_run.sh, C2 batch size only)
    traffic, src = None, None
    if batch == 16 * 65536:      # the 16 x batch of tools/bench_extended.py: its own PMC passes (round 5)
        for tag in ("r05",):
            for suffix in ("", "_before"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_16x{suffix}.json")))["kernels"]
                    key = dom[0] if dom[0] in pmc else dom[0].replace("late", "pipe")
                    traffic, src = pmc[key]["traffic_bytes"], f"profiles/{tag}_pmc_traffic_16x{suffix}.json"
                    break
                except Exception:
                    continue
            if traffic is not None:
                break
    if batch == 65536:
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")))["kernels"]
                key = dom[0] if dom[0] in pmc else dom[0].replace("late", "pipe")
                traffic, src = pmc[key]["traffic_bytes"], f"profiles/{tag}_pmc_traffic.json"
                break
            except Exception:
                continue
    roof = {"bound": "hbm", "kernel": dom[0], "achieved": dom[1]["GB/s"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": dom[1]["GB/s"] / HBM_PEAK_GBPS, "traffic": traffic,
            "traffic_source": (src + ": separate rocprofv3 --pmc passes of this command (2*FETCH_SIZE + WRITE_SIZE)") if src else None,
            "kernels": kern, "keys_per_launch": nt_avg, "unique_rows_per_launch": nu_avg}
    step_bytes = (8 * nt_avg + 8 * (FB + 1) + 16 * nu_avg + nu_avg * D * e + FB * D * o) + \
                 (8 * nt_avg + FB * D * o + 2 * nu_avg * D * e)
    return roof, step_bytes


MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak


def hstu_flops(lengths, heads, dim, causal=True):
    """The reference's FLOP model (examples/commons/utils/perf.py:697-740) without contexts/targets:
    4*H*L^2*d for the two GEMMs, halved by causality; bwd = 2.5 x fwd (hstu_attn_kernel_benchmark.py:343-352)."""
    fl = 0.0
    for L in lengths:
        fl += 4.0 * heads * L * L * dim - (2.0 * heads * L * L * dim if causal else 0.0)
    return fl


def hstu_section(args, device, world, dist=None):
    """Path B on the C3 attention shape (B=32, L=512, H=4, d=256, bf16, causal): fwd + bwd kernels timed with
    HIP events on the launch stream.  Replicas only under N > 1 (attention is per-sequence data parallel)."""
    from hstu import hstu_varlen_bwd, hstu_varlen_fwd

    Bq, L, H, d = args.hstu_batch, args.hstu_seqlen, args.hstu_heads, args.hstu_dim
    T = Bq * L
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(7)
    q, k, v, do = (torch.empty(T, H, d, device=device).uniform_(-1, 1, generator=g).bfloat16() for _ in range(4))
    alpha = 1.0 / d ** 0.5

    def timeit(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    tf = timeit(lambda: hstu_varlen_fwd(q, k, v, cu, L, L, None, None, 1, True, alpha), args.hstu_reps)
    tb = timeit(lambda: hstu_varlen_bwd(do, q, k, v, cu, L, L, None, None, 1, True, alpha), args.hstu_reps)
    if dist is not None:
        t = torch.tensor([tf, tb], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tf, tb = float(t[0]), float(t[1])
    fl = hstu_flops([L] * Bq, H, d)
    tot = (fl + 2.5 * fl) / (tf + tb) / 1e9
    # Both roofs of the op.  Algorithmic bytes: forward reads q, k, v and writes o (4 tensors of T*H*d bf16); backward
    # reads q, k, v, dO and writes dq, dk, dv (7).  At L = 512 the op has 128 FLOP per byte, below the machine balance of
    # 2.5 PFLOP/s / 8 TB/s = 312: C3's attention is HBM-bound before it is MFMA-bound.
    tensor_bytes = T * H * d * 2
    t_hbm_ms = (4 + 7) * tensor_bytes / (HBM_PEAK_GBPS * 1e9) * 1e3
    t_mfma_ms = 3.5 * fl / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e3
    return {"metric": "HSTU seq-tokens/sec (hstu_attn_varlen fwd+bwd kernels)", "value": world * T / (tf + tb) * 1e3,
            "unit": "tokens/s", "fwd_ms": tf, "bwd_ms": tb, "fwd_TFLOPs": fl / tf / 1e9, "bwd_TFLOPs": 2.5 * fl / tb / 1e9,
            "dtype": "bf16 operands, f32 accumulate", "scaling": "replicas only",
            "roofline": {"bound": "mfma", "achieved": tot, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tot / MFMA_BF16_PEAK_TFLOPS},
            "roofline_hbm": {"bound": "hbm", "achieved": (4 + 7) * tensor_bytes / ((tf + tb) * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": (4 + 7) * tensor_bytes / ((tf + tb) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "floor_ms": {"hbm": t_hbm_ms, "mfma": t_mfma_ms},
                             "note": "the binding roof at this shape is the larger floor"},
            "config": {"workload": f"C3 attention: batch {Bq} x L {L} (dense lengths), H {H}, d {d}, causal, alpha 1/sqrt(d)"}}


def hstu_jagged_section(args, device):
    """Path B on the C4 attention shape the retrieval model actually runs: jagged sequences, lengths Zipf(1.2) clipped to
    [32, 4096], 32 sequences, H 4, d 256, causal -- fwd + bwd kernels event-timed, FLOPs by the reference's model."""
    from hstu import hstu_varlen_bwd, hstu_varlen_fwd

    rng = np.random.default_rng(1)
    lengths = np.clip(rng.zipf(1.2, 32) + 31, 32, 4096).astype(np.int64)
    H, d = args.hstu_heads, args.hstu_dim
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int32, device=device)
    T, L = int(cu[-1]), int(lengths.max())
    g = torch.Generator(device=device)
    g.manual_seed(11)
    q, k, v, do = (torch.empty(T, H, d, device=device).uniform_(-1, 1, generator=g).bfloat16() for _ in range(4))
    alpha = 1.0 / d ** 0.5

    def timeit(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    tf = timeit(lambda: hstu_varlen_fwd(q, k, v, cu, L, L, None, None, 1, True, alpha), 8)
    tb = timeit(lambda: hstu_varlen_bwd(do, q, k, v, cu, L, L, None, None, 1, True, alpha), 6)
    fl = hstu_flops([int(x) for x in lengths], H, d)
    tot = 3.5 * fl / (tf + tb) / 1e9
    return {"metric": "HSTU seq-tokens/sec, jagged (C4 attention shape)", "value": T / (tf + tb) * 1e3, "unit": "tokens/s",
            "fwd_ms": tf, "bwd_ms": tb, "fwd_TFLOPs": fl / tf / 1e9, "bwd_TFLOPs": 2.5 * fl / tb / 1e9, "tokens": T,
            "roofline": {"bound": "mfma", "achieved": tot, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tot / MFMA_BF16_PEAK_TFLOPS},
            "config": {"workload": f"C4 attention: 32 jagged sequences, lengths Zipf(1.2) in [32, 4096] (max {L}, {T} tokens), "
                                   f"H {H}, d {d}, causal"}}


def hstu_long_section(args, device):
    """Path B in its compute-bound regime: 8 dense sequences of 4096 (H 4, d 256, causal) -- the shape the round-3 review set
    its forward / backward targets on (>= 900 / 750 TFLOP/s).  `sustained_peak` is what tools/ubench_mfma.hip measures for
    back-to-back 32x32x16 bf16 MFMAs on all 256 CUs (the clock drops from 2.4 to ~1.6-1.9 GHz): profiles/r04_ubench_mfma.txt."""
    from hstu import hstu_varlen_bwd, hstu_varlen_fwd

    Bq, L, H, d = 8, 4096, args.hstu_heads, args.hstu_dim
    T = Bq * L
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(13)
    q, k, v, do = (torch.empty(T, H, d, device=device).uniform_(-1, 1, generator=g).bfloat16() for _ in range(4))
    alpha = 1.0 / d ** 0.5

    def timeit(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    tf = timeit(lambda: hstu_varlen_fwd(q, k, v, cu, L, L, None, None, 1, True, alpha), 10)
    tb = timeit(lambda: hstu_varlen_bwd(do, q, k, v, cu, L, L, None, None, 1, True, alpha), 6)
    fl = hstu_flops([L] * Bq, H, d)
    tot = 3.5 * fl / (tf + tb) / 1e9
    return {"metric": "HSTU attention, dense 8 x 4096", "value": T / (tf + tb) * 1e3, "unit": "tokens/s", "fwd_ms": tf, "bwd_ms": tb,
            "fwd_TFLOPs": fl / tf / 1e9, "bwd_TFLOPs": 2.5 * fl / tb / 1e9,
            "roofline": {"bound": "mfma", "achieved": tot, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tot / MFMA_BF16_PEAK_TFLOPS,
                         "fwd_frac": fl / tf / 1e9 / MFMA_BF16_PEAK_TFLOPS, "sustained_peak": 1900.0,
                         "sustained_peak_source": "tools/ubench_mfma.hip, profiles/r04_ubench_mfma.txt"},
            "config": {"workload": f"attention: batch {Bq} x L {L} (dense lengths), H {H}, d {d}, causal, alpha 1/sqrt(d)"}}


def model_shapes_section(args, device):
    """The embedding step at the shapes an HSTU model's embedding collection sends (the round-3 review's item 5):
    eight tables of 6.25 M rows, (i) pooled SUM, 8 x 8192 bags of ~5.5 keys, (ii) sequence lookups, 8 x 16 384 tokens --
    fwd + bwd (SGD) per step and the eval forward, Zipf(--alpha) keys per table.  Both take the CSR-writing partition path with
    table-aligned partitions since round 4 (tools/bench_model_shapes.py has the other shapes and the A/B switches)."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    T, rows = 8, 6_250_000
    res = {}
    for name, pooling, bags, hot in (("pooled_8x8192_bags", "SUM", 8192, 5), ("sequence_8x16384_tokens", "NONE", 16384, 1)):
        opts = [DynamicEmbTableOptions(dim=args.dim, max_capacity=rows, embedding_dtype=torch.float32, index_type=torch.int64,
                                       score_strategy=DynamicEmbScoreStrategy.TIMESTAMP,
                                       initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM, lower=-0.01, upper=0.01))
                for _ in range(T)]
        m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=list(range(T)), pooling_mode=getattr(DynamicEmbPoolingMode, pooling),
                                            output_dtype=torch.bfloat16, optimizer=EmbOptimType.SGD, learning_rate=0.1, device=device)
        m.train()
        g = torch.Generator(device=device)
        g.manual_seed(77)
        w = torch.arange(1, rows + 1, device=device, dtype=torch.float64).pow_(-args.alpha)
        cdf = torch.cumsum(w, 0)
        cdf /= cdf[-1].clone()
        batches = []
        for _ in range(6):
            lens = (torch.ones(T * bags, dtype=torch.int64, device=device) if hot == 1
                    else torch.randint(1, 2 * hot, (T * bags,), device=device, generator=g))
            off = torch.zeros(T * bags + 1, dtype=torch.int64, device=device)
            off[1:] = torch.cumsum(lens, 0)
            u = torch.rand(int(off[-1].item()), device=device, dtype=torch.float64, generator=g)
            ranks = torch.searchsorted(cdf, u).clamp_(max=rows - 1)
            batches.append(((ranks * 2654435761 % rows).contiguous(), off))
        del w, cdf
        with torch.no_grad():
            for k, o in batches:
                m._forward_impl(k, o, train=True)
        nus = []
        for k, o in batches:     # unique rows per step (for the step's algorithmic bytes)
            out, st = m._forward_impl(k, o, train=True)
            nus.append(int(st.uoff[-1].item()))
            grad = (torch.randn_like(out.float()) * 0.01).to(out.dtype)
            m._backward_impl(st, grad)
        out, st = m._forward_impl(*batches[0], train=True)
        grad = (torch.randn_like(out.float()) * 0.01).to(out.dtype)
        m._backward_impl(st, grad)

        def step(i):
            k, o = batches[i % len(batches)]
            out, st = m._forward_impl(k, o, train=True)
            m._backward_impl(st, grad if out.shape == grad.shape else torch.zeros_like(out))

        def ev(i):
            k, o = batches[i % len(batches)]
            m._forward_impl(k, o, train=False)

        def timeit(fn, reps=60):
            for i in range(8):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        ms = timeit(step)
        m.eval()
        with torch.no_grad():
            ms_eval = timeit(ev)
        nk = sum(k.numel() for k, _ in batches) / len(batches)
        nu = float(np.mean(nus))
        D, e, o = args.dim, 4, 2
        rows_out = T * bags if hot != 1 else nk       # output rows: one per bag (pooled) or per key (sequence)
        # the byte model of the C2 step (kernel_roofline above): index words per key / bag / unique row, every unique row read
        # once in the forward and read + written in the backward, the output written, its gradient read
        step_bytes = (8 * nk + 8 * (T * bags + 1) + 16 * nu + nu * D * e + rows_out * D * o) + (8 * nk + rows_out * D * o + 2 * nu * D * e)
        gbps = step_bytes / ms / 1e6
        res[name] = {"ms_per_step": ms, "eval_forward_ms": ms_eval, "keys_per_step": nk, "unique_rows_per_step": nu,
                     "lookups_per_s": nk / ms * 1e3, "path_c": bool(getattr(st, "lazy", False)),
                     "step_roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": gbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": step_bytes}}
        del m, batches
        torch.cuda.empty_cache()
    res["config"] = {"workload": f"{T} tables x {rows} rows, dim {args.dim}, fp32 rows, SGD, Zipf({args.alpha}) keys per table"}
    return res


def c2_16x_section(args, module, device):
    """The bandwidth-regime figure SURVEY 8(d) asks for: the C2 step at 16 x the batch (B = 1,048,576 bags, ~5.8 M keys) on
    the same table: step time and the step-level roofline (minimal bytes / time)."""
    B16 = 16 * args.batch
    batches = zipf_batches(args.rows, args.alpha, B16, 3, device, seed=777)
    grad = (torch.randn(B16, args.dim, device=device) * 0.01).to(torch.bfloat16)
    with torch.no_grad():
        for keys, offsets in batches:
            module._forward_impl(keys, offsets, train=True)
    nu = []
    for keys, offsets in batches:   # warm-up (and the unique counts)
        out, st = module._forward_impl(keys, offsets, train=True)
        nu.append(int(st.uoff[-1].item()))
        module._backward_impl(st, grad)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        for keys, offsets in batches:
            out, st = module._forward_impl(keys, offsets, train=True)
            module._backward_impl(st, grad)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (reps * len(batches)) * 1e3
    nt = float(np.mean([k.numel() for k, _ in batches]))
    nua = float(np.mean(nu))
    D, e, o = args.dim, 4, 2
    step_bytes = (8 * nt + 8 * (B16 + 1) + 16 * nua + nua * D * e + B16 * D * o) + (8 * nt + B16 * D * o + 2 * nua * D * e)
    gbps = step_bytes / ms / 1e6
    del batches, grad
    torch.cuda.empty_cache()
    return {"ms_per_step": ms, "keys_per_step": nt, "unique_rows_per_step": nua, "lookups_per_s": nt / ms * 1e3,
            "step_roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                              "algorithmic_bytes_per_step": step_bytes}}


def c2_filling_section(args, device, fill=300, window=40):
    """The C2 step while the table FILLS UP: `fill` distinct batches into an empty table -- at 300 the 10 M-row table is ~3/4 full and a
    few new keys per step meet a full bucket, so the partition kernel evicts for them -- and the time of the last `window` of them.
    Every batch here is new (inserts in every step), unlike the headline's steady state over pre-inserted batches; found in round 5
    (`bench.py --steps 300`, profiles/r05_eviction_regime.txt): the regime a long training run with a full table lives in."""
    batches = zipf_batches(args.rows, args.alpha, args.batch, fill, device, seed=4321)
    m = build_module(args.rows, args.dim, device)
    m.train()
    grad = (torch.randn(args.batch, args.dim, device=device) * 0.01).to(torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i, (keys, offsets) in enumerate(batches):
        if i == fill - window:
            e0.record()
        out, st = m._forward_impl(keys, offsets, train=True)
        m._backward_impl(st, grad)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / window
    size = int(m.size())
    nt = float(np.mean([k.numel() for k, _ in batches[fill - window:]]))
    del batches, m, grad
    torch.cuda.empty_cache()
    return {"ms_per_step": ms, "lookups_per_s": nt / ms * 1e3, "distinct_batches": fill, "timed_last": window, "table_rows": size,
            "load": size / args.rows, "note": "every batch new (inserts each step; full buckets evict)"}


def sharded_stage_times(sharded, batches, grad, dist, device, world, step_ms):
    """Where a sharded step's time goes, so that a scaling curve explains itself: the four stages of one step run back to
    back WITHOUT overlap (plain schedule), each bracketed by HIP events on the launch stream -- input_dist (bucketize + the
    two all-to-alls of lengths and keys), lookup (local fused forward), output_dist (all-to-all of partial sums + their sum),
    backward (all-gather of the gradients + local reduce / SGD) -- max over ranks; `exposed_comm` = the overlapped step time
    of the timed window minus the two compute stages, i.e. the communication (and host glue) the overlap did not hide.
    Only the partial-sum dist has this stage structure (the rows dist reports its forward / backward halves)."""
    impl = sharded.impl
    n = min(len(batches), 10)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    acc = {}

    def timed(name, fn):
        e0, e1 = ev(), ev()
        e0.record()
        r = fn()
        e1.record()
        acc.setdefault(name, []).append((e0, e1))
        return r

    staged = hasattr(impl, "dist_input")
    for keys, offsets in batches[:n]:
        if staged:
            sk = timed("input_dist", lambda: impl.dist_input(keys, offsets))
            out_local, lctx = timed("lookup", lambda: impl.lookup(sk, True))
            timed("output_dist", lambda: impl.dist_output(sk, out_local))
            g_all = timed("backward_comm", lambda: impl.dist_grads(sk, grad))
            timed("backward_local", lambda: impl.local.backward(lctx, g_all))
        else:
            out, ctx = timed("forward", lambda: impl.forward(keys, offsets, True))
            timed("backward", lambda: impl.backward(ctx, grad))
    torch.cuda.synchronize()
    names = sorted(acc)
    t = torch.tensor([float(np.median([a.elapsed_time(b) for a, b in acc[k]])) for k in names], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = {k: float(v) for k, v in zip(names, t.tolist())}
    if staged:
        out["backward"] = out["backward_comm"] + out["backward_local"]
        out["exposed_comm"] = max(0.0, step_ms - out["lookup"] - out["backward_local"])
    out["mode"] = getattr(sharded, "mode", "?")
    out["note"] = "stages timed one after the other (no overlap), median of %d steps, max over ranks; the timed window overlaps them" % n
    return out


def _flush_c_stdout():
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _self_launch(args):
    """`python bench.py --gpus N` outside a launcher: re-exec under torch.distributed.run (one rank per GPU over RCCL), the
    command the driver uses for N > 1; rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    sharded_path = world > 1 or args.force_sharded
    if sharded_path:
        import torch.distributed as dist

        if world == 1 and "MASTER_PORT" not in os.environ:    # --force-sharded on one GPU: no TCP port to collide on
            import tempfile, uuid
            dist.init_process_group("nccl", init_method="file://" + os.path.join(tempfile.gettempdir(), "mi355_bench_pg_" + uuid.uuid4().hex),
                                    rank=0, world_size=1, device_id=device)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import dynamicemb_extensions as ext

    n_batches = args.steps + args.warmup
    batches = zipf_batches(args.rows, args.alpha, args.batch, n_batches, device, seed=1234 + rank)

    if not sharded_path:
        module = build_module(args.rows, args.dim, device)
        module.train()

        # round 5: the timed step is the module's PUBLIC path -- module(keys, offsets) -> _LookupFunction.apply -> autograd
        # backward -- which is what a TorchRec caller drives (round-4 review item 4); the direct _forward_impl / _backward_impl
        # step is recorded next to it as `step_via_impl_ms`
        def fwd(keys, offsets):
            return module(keys, offsets), None

        def bwd(out, grad):
            out.backward(grad)
    else:
        from dynamicemb.sharded import ShardedPooledLookup

        sharded = ShardedPooledLookup(args.rows, args.dim, device, world, rank, mode=args.shard_mode,
                                      keys_per_step=int(args.batch * 5.5), batch=args.batch,
                                      capacity_factor=args.capacity_factor or None)

        def fwd(keys, offsets, nxt=None):
            return sharded.forward(keys, offsets, next_batch=nxt)

        def bwd(st, grad):
            sharded.backward(st, grad)

    grad = (torch.randn(args.batch, args.dim, device=device) * 0.01).to(torch.bfloat16)

    # steady state: every key of every batch is already in the table
    with torch.no_grad():
        for keys, offsets in batches:
            if sharded_path:
                fwd(keys, offsets)
            else:
                module._forward_impl(keys, offsets, train=True)   # (insert every key: module() under no_grad is the eval forward)
    torch.cuda.synchronize()

    def step(i):
        keys, offsets = batches[i]
        if sharded_path:   # the next batch's key exchange goes out under this batch's lookup / backward (exchange stream)
            out, st = fwd(keys, offsets, batches[i + 1] if i + 1 < n_batches else None)
        else:
            out, st = fwd(keys, offsets)
        bwd(out if not sharded_path else st, grad)
        return out, st

    for i in range(args.warmup):
        step(i)
    if sharded_path:
        dist.barrier()
        _flush_c_stdout()   # every rank: RCCL's banner (NCCL_DEBUG=VERSION on the box) leaves the C buffer now, not at exit
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_batches):
        step(i)
    torch.cuda.synchronize()
    if sharded_path:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    keys_local = sum(batches[i][0].numel() for i in range(args.warmup, n_batches))
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([keys_local], device=device, dtype=torch.float64)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        keys_total = float(k.item())
    else:
        keys_total = float(keys_local)

    result = {
        "metric": "embedding lookups/sec (DynamicEmb lookup+pool fwd+bwd, steady state)",
        "value": keys_total / elapsed,
        "unit": "lookups/s",
        "n_gpus": world,
        "ranks": (dist.get_world_size() if sharded_path else 1),   # what the process group (RCCL) actually holds
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 table rows, f32 accumulate, bf16 pooled output / gradients",
        "data": "synthetic",
        "config": {"workload": f"C2: DynamicEmb 1 table x {args.rows} rows x {args.dim}-D fp32, Zipf-{args.alpha} keys, "
                               f"batch {args.batch} bags x randint(1,11) keys, SUM pooling, SGD, lookup+pool fwd/bwd",
                   "keys_per_step": keys_total / args.steps / world, "parallelism": f"row-wise mp{world}" if world > 1 else "1 GPU"},
    }

    # the K timed steps above are a few milliseconds of GPU time at the driver's K = 20: a second, longer window (cycling
    # over the same batches until >= 0.25 s) shows what the rate is once clocks and caches have settled
    sus_steps, t1 = 0, time.perf_counter()
    while True:
        for i in range(args.warmup, n_batches):
            step(i)
        sus_steps += args.steps
        torch.cuda.synchronize()
        spent = time.perf_counter() - t1
        if world > 1:   # ONE decision for all ranks (each reading its own clock would leave a rank alone in a collective)
            t = torch.tensor([spent], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            spent = float(t.item())
        if spent >= 0.25 or sus_steps >= 200 * args.steps:
            break
    sus = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([sus], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sus = float(t.item())
    result["sustained"] = {"steps": sus_steps, "seconds": sus, "ms_per_step": 1e3 * sus / sus_steps,
                           "value": keys_total / args.steps * sus_steps / sus, "unit": "lookups/s"}

    if sharded_path:
        # which exchange moved the bytes of the timed steps: "native" = the library's own RCCL calls (csrc/exchange.hip: created,
        # agreed on by all ranks and, at W > 1, checked once against the c10d sequence on the first batch), "c10d" = the
        # torch.distributed call sequence (the fallback every failure of the former lands on)
        lk = getattr(sharded.impl, "inner", sharded.impl)
        result["exchange"] = lk.exchange
        result["exchange_selfchecked"] = bool(getattr(lk, "exchange_selfchecked", False))
        try:
            result["stages_ms"] = sharded_stage_times(sharded, batches[args.warmup:], grad, dist, device, world,
                                                      result["sustained"]["ms_per_step"])
        except Exception as e:      # noqa: BLE001 -- a diagnostic must not cost the headline line
            result["stages_ms"] = {"error": repr(e)}

    if rank == 0 and not sharded_path:
        # the same step through the module's internal entry points (_forward_impl / _backward_impl: no autograd function, no
        # engine hop) -- what rounds 1-4 timed as the headline; the public path above must stay within a few percent of it
        def step_impl(i):
            keys, offsets = batches[i]
            out, st = module._forward_impl(keys, offsets, train=True)
            module._backward_impl(st, grad)

        for i in range(args.warmup):
            step_impl(i)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        n_impl = 0
        while True:
            for i in range(args.warmup, n_batches):
                step_impl(i)
            n_impl += args.steps
            torch.cuda.synchronize()
            if time.perf_counter() - ta >= 0.25 or n_impl >= 200 * args.steps:
                break
        result["step_via_impl_ms"] = 1e3 * (time.perf_counter() - ta) / n_impl
        result["step_via_autograd_ms"] = result["sustained"]["ms_per_step"]
        result["step_via_autograd_note"] = ("the timed region IS module.forward(keys, offsets) + out.backward(grad) since round 5; "
                                            "step_via_impl_ms = the same batches through _forward_impl / _backward_impl")

    if rank == 0 and not sharded_path and not args.no_extra:
        # the reference's prefetch-pipeline order on the same module and batches (round 6: prefetch_async runs the index stage of
        # batch k + 1 on the partitioned path under the backward of batch k).  NOT the headline: on one GPU without a dense model
        # the latency-bound index kernels and the bandwidth-bound backward slow each other down (profiles/r06_pipelined_timeline.txt)
        try:
            result["pipelined_ms_per_step"] = pipelined_ms(module, batches[args.warmup:], grad, max(args.steps, 100))
            result["pipelined_note"] = ("prefetch_async(batch k+1) before backward(batch k), public calls; serial headline above. "
                                        "Pinning prefetch of rounds 1-5 on the same loop: MI355_PREFETCH_C=0")
        except Exception as e:      # noqa: BLE001
            result["pipelined_ms_per_step"] = None
            result["pipelined_note"] = repr(e)

    if rank == 0 and not sharded_path and not args.no_kernel_timing:
        roof, step_bytes = kernel_roofline(module, batches[args.warmup:], grad, args.batch, args.dim)
        result["roofline"] = roof
        gbps = step_bytes / (sus / sus_steps) / 1e9
        result["step_algorithmic_GBps"] = gbps
        result["step_roofline"] = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": gbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": step_bytes,
                                   "note": "whole fwd+bwd step (SURVEY 8(d) minimal bytes) over the sustained window"}

    if rank == 0 and not sharded_path and not args.no_kernel_timing and not args.no_extra:
        result["c2_16x"] = c2_16x_section(args, module, device)
        result["model_shapes"] = model_shapes_section(args, device)
        result["c2_table_filling"] = c2_filling_section(args, device)

    if not args.no_hstu:
        try:
            h = hstu_section(args, device, world, dist if sharded_path else None)
        except Exception as e:      # noqa: BLE001
            if world == 1:
                raise
            h = {"error": repr(e)}
        if rank == 0:
            result["hstu"] = h
        if rank == 0 and world == 1 and not args.no_extra:
            result["hstu_jagged"] = hstu_jagged_section(args, device)
            result["hstu_l4096"] = hstu_long_section(args, device)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # keys index host rows directly (the permuted keys are already in [0, rows))
        bc = [(k.cpu(), o.cpu()) for k, o in batches[: min(len(batches), 4)]]
        result["cpu_baseline"] = cpu_baseline(args, bc)

    if sharded_path:
        dist.destroy_process_group()
    if rank == 0:
        # the box exports NCCL_DEBUG=VERSION: RCCL writes a five-line banner to the C stdout buffer; push it out first so
        # that the JSON line is the LAST line of stdout
        _flush_c_stdout()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
