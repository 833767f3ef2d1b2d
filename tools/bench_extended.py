#!/usr/bin/env python
"""Secondary measurements SURVEY.md 8(d) asks to report next to the headline bench.py line (one GPU):

  path A  C2 forward only (train / eval), Adam, bf16 table rows, a 16x batch (B = 1,048,576: the bandwidth term
          dominates the launch chain), hotness-1 sequence lookups, a cold insert-heavy pass over an empty table,
          a hybrid HBM + pinned-host table (C4 layout, scaled to what a short run can populate)
  path B  jagged attention with Zipf(1.2) lengths clipped to [32,512] (C3) and [32,4096] (C4), and the C5 decode step
          (paged KV cache of 3968 tokens + 128 new history + 256 candidates per sequence, append_kvcache + attention)

Writes ONE JSON object to stdout (and to --out).  Everything is timed with HIP events on the launch stream with the
inputs resident in HBM, like bench.py.  Not part of the driver's contract -- the numbers are copied into profiles/.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "recsys-examples_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import bench

DEV = torch.device("cuda", 0)


def timed(fn, n, warm=2):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(warm, warm + n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def make_module(rows, dim, optimizer="SGD", table_dtype=torch.float32, pooling="SUM", out_dtype=torch.bfloat16,
                local_hbm=0, storage_mode=None):
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    opt = DynamicEmbTableOptions(dim=dim, max_capacity=rows, embedding_dtype=table_dtype, index_type=torch.int64,
                                 score_strategy=DynamicEmbScoreStrategy.TIMESTAMP, local_hbm_for_values=local_hbm,
                                 initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM,
                                                                            lower=-0.01, upper=0.01))
    m = BatchedDynamicEmbeddingTablesV2([opt], feature_table_map=[0], pooling_mode=getattr(DynamicEmbPoolingMode, pooling),
                                        output_dtype=out_dtype, optimizer=getattr(EmbOptimType, optimizer),
                                        learning_rate=0.1, device=DEV, storage_mode=storage_mode)
    m.train()
    return m


def run_table_case(name, rows, dim, batch, n_steps, fwd_only=False, evaluate=False, preinsert=True, hotness1=False, **mk):
    """-> dict(ms_per_step, lookups_per_s, keys_per_step)"""
    n_b = n_steps + 2
    if hotness1:
        g = torch.Generator(device=DEV)
        g.manual_seed(5)
        batches = []
        full = bench.zipf_batches(rows, 0.99, batch, n_b, DEV, seed=99)
        for keys, _ in full:
            batches.append((keys[:batch].contiguous(), torch.arange(batch + 1, dtype=torch.int64, device=DEV)))
    else:
        batches = bench.zipf_batches(rows, 0.99, batch, n_b, DEV, seed=4321)
    m = make_module(rows, dim, **mk)
    if preinsert:
        with torch.no_grad():
            for keys, offsets in batches:
                m._forward_impl(keys, offsets, train=True)
    if evaluate:
        m.eval()
    n_out = batches[0][0].numel() if mk.get("pooling") == "NONE" else batch
    grads = None

    def step(i):
        nonlocal grads
        keys, offsets = batches[i]
        out, st = m._forward_impl(keys, offsets, train=not evaluate)
        if not (fwd_only or evaluate):
            if grads is None or grads.size(0) != out.size(0):
                grads = (torch.randn(out.size(0), dim, device=DEV) * 0.01).to(out.dtype)
            m._backward_impl(st, grads)

    if mk.get("pooling") == "NONE" and not hotness1:
        raise ValueError("sequence mode is benchmarked with hotness 1")
    ms = timed(step, n_steps, warm=2 if preinsert else 0)
    nk = float(np.mean([b[0].numel() for b in batches[(2 if preinsert else 0):(2 if preinsert else 0) + n_steps]]))
    res = {"ms_per_step": ms, "lookups_per_s": nk / ms * 1e3, "keys_per_step": nk, "bags": batch, "rows": rows}
    if name == "c2_16x_batch":
        grad = (torch.randn(batch, dim, device=DEV) * 0.01).to(torch.bfloat16)
        roof, step_bytes = bench.kernel_roofline(m, batches[2:2 + min(n_steps, 4)], grad, batch, dim)
        res["roofline"] = roof
        res["step_algorithmic_GBps"] = step_bytes / ms / 1e6
    del m
    torch.cuda.empty_cache()
    return res


def zipf_lengths(n, a, lo, hi, seed):
    rng = np.random.default_rng(seed)
    return np.clip(rng.zipf(a, n) + lo - 1, lo, hi).astype(np.int64)


def hstu_jagged(lengths, H, d, reps=10):
    from hstu import hstu_varlen_bwd, hstu_varlen_fwd

    cu = torch.tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int32, device=DEV)
    T, L = int(cu[-1]), int(max(lengths))
    q, k, v, do = (torch.empty(T, H, d, device=DEV).uniform_(-1, 1).bfloat16() for _ in range(4))
    alpha = 1.0 / d ** 0.5
    tf = timed(lambda i: hstu_varlen_fwd(q, k, v, cu, L, L, None, None, 1, True, alpha), reps)
    tb = timed(lambda i: hstu_varlen_bwd(do, q, k, v, cu, L, L, None, None, 1, True, alpha), reps)
    fl = bench.hstu_flops([int(x) for x in lengths], H, d)
    return {"tokens": T, "max_len": L, "fwd_ms": tf, "bwd_ms": tb, "tokens_per_s_fwd_bwd": T / (tf + tb) * 1e3,
            "fwd_TFLOPs": fl / tf / 1e9, "bwd_TFLOPs": 2.5 * fl / tb / 1e9}


def hstu_decode(B=8, H=4, d=256, P=32, old=3968, new_hist=128, cand=256, reps=10):
    """C5: per sequence `old` cached tokens, `new_hist` new history tokens appended this step, `cand` candidates."""
    from hstu import append_kvcache, hstu_attn_varlen_func

    qlen, cachelen = new_hist + cand, old + new_hist
    klen = cachelen + cand
    ti = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=DEV)
    npg = (cachelen + P - 1) // P
    rng = np.random.default_rng(3)
    page_ids = ti(rng.permutation(B * npg))
    page_off = ti(np.arange(B + 1) * npg)
    last = ti(np.full(B, cachelen - (npg - 1) * P))
    cuq, cuk, tgt = ti(np.arange(B + 1) * qlen), ti(np.arange(B + 1) * klen), ti(np.full(B, cand))
    bidx = ti(np.repeat(np.arange(B), new_hist))
    pos = ti(np.tile(old + np.arange(new_hist), B))
    cand_off = ti(np.arange(B + 1) * cand)
    nnz = ti([B * new_hist])
    cache = (torch.randn(B * npg, 2, P, H, d, device=DEV) * 0.5).bfloat16()
    q, k, v = (torch.empty(B * qlen, H, d, device=DEV).uniform_(-1, 1).bfloat16() for _ in range(3))

    def step(i):
        append_kvcache(k, v, bidx, pos, cand_off, nnz, 0, cache, page_ids, page_off, last, 0)
        hstu_attn_varlen_func(q, k, v, cuq, cuk, None, None, qlen, klen, float(klen), None, tgt, window_size=(-1, 0),
                              alpha=1.0 / d ** 0.5, kv_cache=cache, page_offsets=page_off, page_ids=page_ids,
                              last_page_lens=last)

    ms = timed(step, reps)
    # history rows see the cached prefix causally, candidates see all history + themselves
    fl = 0.0
    for r in range(new_hist):
        fl += 4.0 * H * d * (old + r + 1)
    fl += 4.0 * H * d * cand * (cachelen + 1)
    fl *= B
    return {"batch": B, "cached": old, "new_history": new_hist, "candidates": cand, "page_size": P, "heads": H, "dim": d,
            "ms_per_step": ms, "query_tokens_per_s": B * qlen / ms * 1e3, "TFLOPs": fl / ms / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--only", default=None, help="comma-separated case names")
    args = ap.parse_args()
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    R, D, B = args.rows, 128, 65536
    cases = {
        "c2_fwd_bwd_sgd": lambda: run_table_case("c2", R, D, B, args.steps),
        "c2_fwd_only_train": lambda: run_table_case("c2f", R, D, B, args.steps, fwd_only=True),
        "c2_fwd_only_eval": lambda: run_table_case("c2e", R, D, B, args.steps, evaluate=True),
        "c2_adam": lambda: run_table_case("c2a", R, D, B, args.steps, optimizer="ADAM"),
        "c2_bf16_table": lambda: run_table_case("c2h", R, D, B, args.steps, table_dtype=torch.bfloat16),
        "c2_16x_batch": lambda: run_table_case("c2_16x_batch", R, D, 16 * B, 6),
        "c2_hotness1_sequence": lambda: run_table_case("c2s", R, D, B, args.steps, hotness1=True, pooling="NONE"),
        "c2_cold_insert": lambda: run_table_case("c2c", R, D, B, 10, preinsert=False),
        "c4_hybrid_hbm_plus_host": lambda: run_table_case("c4", R, D, B, 10, local_hbm=(R // 8) * D * 4),
        "hstu_jagged_zipf_32_512": lambda: hstu_jagged(zipf_lengths(32, 1.2, 32, 512, 0), 4, 256),
        "hstu_jagged_zipf_32_4096": lambda: hstu_jagged(zipf_lengths(32, 1.2, 32, 4096, 1), 4, 256),
        "hstu_dense_32x4096": lambda: hstu_jagged(np.full(32, 4096), 4, 256, reps=4),
        "c5_paged_decode": hstu_decode,
    }
    only = set(args.only.split(",")) if args.only else None
    out = {"device": torch.cuda.get_device_name(0)}
    for name, fn in cases.items():
        if only and name not in only:
            continue
        try:
            out[name] = fn()
        except Exception as e:  # a case that is not supported is reported, not hidden
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    js = json.dumps(out)
    print(js)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
