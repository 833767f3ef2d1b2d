"""The embedding step at the shapes the HSTU models send (examples/hstu: an EmbeddingCollection -- SEQUENCE embeddings,
several tables per rank), next to C2's single pooled table: fwd + bwd (SGD) per step and the kernels of each.
    python tools/bench_model_shapes.py [--steps 50]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                          DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--eval", action="store_true", help="time the eval / inference forward of every shape instead of the training step")
ap.add_argument("--only", type=int, default=-1, help="run only case N (0-based; for a rocprofv3 kernel trace of one shape)")
a = ap.parse_args()
dev = torch.device("cuda")


def module(T, rows, pooling):
    opts = [DynamicEmbTableOptions(dim=128, max_capacity=rows, embedding_dtype=torch.float32, index_type=torch.int64,
                                   score_strategy=DynamicEmbScoreStrategy.TIMESTAMP,
                                   initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM, lower=-0.01, upper=0.01))
            for _ in range(T)]
    m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=list(range(T)), pooling_mode=getattr(DynamicEmbPoolingMode, pooling),
                                        output_dtype=torch.bfloat16, optimizer=EmbOptimType.SGD, learning_rate=0.1, device=dev)
    m.train()
    return m


def zipf_keys(rng, n, rows):
    u = rng.random(n)
    r = np.minimum((np.exp(u * np.log(rows)) - 1).astype(np.int64), rows - 1)      # p(r) ~ 1 / r
    return (r * 2654435761 % rows).astype(np.int64)


def case(name, T, rows, tokens, pooling, hot):
    """T tables (one feature each), `tokens` bags per feature, `hot` keys per bag (1 = sequence lookups)"""
    rng = np.random.default_rng(len(name))
    m = module(T, rows, pooling)
    batches = []
    for _ in range(6):
        lens = np.full(T * tokens, hot, np.int64) if hot == 1 else rng.integers(1, 2 * hot, T * tokens)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        keys = zipf_keys(rng, int(off[-1]), rows)
        batches.append((torch.from_numpy(keys).to(dev), torch.from_numpy(off).to(dev)))
    with torch.no_grad():
        for k, o in batches:
            m._forward_impl(k, o, train=True)
    out, st = m._forward_impl(*batches[0], train=True)
    g = (torch.randn_like(out.float()) * 0.01).to(out.dtype)
    m._backward_impl(st, g)

    def step(i):
        k, o = batches[i % len(batches)]
        out, st = m._forward_impl(k, o, train=True)
        gg = g if out.shape == g.shape else (torch.zeros_like(out))
        m._backward_impl(st, gg)

    if a.eval:
        m.eval()

        def step(i):   # noqa: F811
            k, o = batches[i % len(batches)]
            with torch.no_grad():
                m._forward_impl(k, o, train=False)

    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    nk = float(np.mean([k.numel() for k, _ in batches]))
    print(f"{name:46s} keys/step {nk:9.0f}  {ms:7.4f} ms/step  {nk / ms / 1e6:7.2f} G lookups/s", flush=True)


CASES = [("C2: 1 table, pooled SUM, 65536 bags x ~5.5", 1, 10_000_000, 65536, "SUM", 5),
         ("1 table, sequence, 131072 tokens", 1, 10_000_000, 131072, "NONE", 1),
         ("8 tables, sequence, 8 x 16384 tokens", 8, 6_250_000, 16384, "NONE", 1),
         ("8 tables, pooled SUM, 8 x 8192 bags x ~5.5", 8, 6_250_000, 8192, "SUM", 5),
         ("1 table, sequence, 16384 tokens (one feature)", 1, 10_000_000, 16384, "NONE", 1)]
for i, c in enumerate(CASES):
    if a.only < 0 or a.only == i:
        case(*c)
