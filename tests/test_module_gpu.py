"""Module-level GPU tests of BatchedDynamicEmbeddingTablesV2 (MI355X build), mirroring the reference's
corelib/dynamicemb/test/test_batched_dynamic_embedding_tables_v2.py: the 11-key / 4-feature /
batch-2 fixture with its train == eval / zero-for-unknown / insert-on-train assertions
(test_forward_train_eval :1436-1591), the DEBUG initializer known answer (test/unit_tests/debug.py
:157-224) and the 10-iteration optimizer twin (test_backward :1636-1746; the twin here is a dense
torch.nn.Embedding + torch.optim, FBGEMM TBE being unavailable)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mods():
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode,
                                              DynamicEmbPoolingMode, DynamicEmbScoreStrategy, DynamicEmbTableOptions,
                                              EmbOptimType)

    return (BatchedDynamicEmbeddingTablesV2, DynamicEmbInitializerArgs, DynamicEmbInitializerMode,
            DynamicEmbPoolingMode, DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)


INDICES = [0, 1, 12, 64, 8, 12, 15, 2, 7, 105, 0]
OFFSETS = [0, 2, 3, 5, 6, 8, 10, 10, 11]


def _make(dims, pooling, opt="SGD", init_mode=None, strategy=None, out_dtype=torch.float32, cap=2048, **kw):
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    init = IA(mode=init_mode or IM.UNIFORM, lower=-0.5, upper=0.5)
    opts = [TO(dim=d, max_capacity=cap, index_type=torch.int64, embedding_dtype=torch.float32, initializer_args=init,
               score_strategy=strategy if strategy is not None else SS.TIMESTAMP) for d in dims]
    return B2(table_options=opts, table_names=[f"table{i}" for i in range(len(dims))], feature_table_map=[0, 0, 1, 2],
              pooling_mode=getattr(PM, pooling), optimizer=getattr(OT, opt), output_dtype=out_dtype,
              device=torch.device(DEV), **kw)


@pytest.mark.parametrize("pooling", ["NONE", "SUM", "MEAN"])
@pytest.mark.parametrize("dims", [[8, 8, 8], [7, 7, 7], [8, 16, 32], [7, 11, 13]])
@pytest.mark.parametrize("opt,params", [("SGD", dict(learning_rate=0.3)),
                                        ("ADAM", dict(learning_rate=0.3, weight_decay=0.06, eps=3e-5, beta1=0.8, beta2=0.888)),
                                        ("EXACT_ROWWISE_ADAGRAD", dict(learning_rate=0.3, eps=3e-5))])
def test_forward_train_eval(pooling, dims, opt, params):
    if pooling == "NONE" and len(set(dims)) > 1:
        pytest.skip("sequence mode requires uniform dims (as the reference)")
    m = _make(dims, pooling, opt, **params)
    idx = torch.tensor(INDICES, dtype=torch.int64, device=DEV)
    off = torch.tensor(OFFSETS, dtype=torch.int64, device=DEV)
    B = 2
    emb_train = m(idx, off)
    if pooling == "NONE":
        assert emb_train.shape == (len(INDICES), dims[0])
    else:
        assert emb_train.shape == (B, sum(dims[t] for t in [0, 0, 1, 2]))
    with torch.no_grad():
        m.eval()
        emb_eval = m(idx, off)
        m(idx + 1024, off)  # all keys missing in eval
    torch.testing.assert_close(emb_train, emb_eval)
    idx_ne = torch.tensor([777] + INDICES[1:], dtype=torch.int64, device=DEV)
    emb_ne = m(idx_ne, off)
    m.train()
    emb_tne = m(idx_ne, off)
    if pooling == "NONE":
        torch.testing.assert_close(emb_train[1:], emb_ne[1:])
        assert torch.all(emb_ne[0] == 0)
        assert torch.all(emb_tne[0] != 0)
        torch.testing.assert_close(emb_tne[1:], emb_ne[1:])
    else:
        torch.testing.assert_close(emb_ne[1], emb_train[1])
        torch.testing.assert_close(emb_tne[1], emb_ne[1])
        assert not torch.allclose(emb_ne[0], emb_train[0])
    # 10 distinct keys of the fixture + key 777
    assert int(m.size().item()) == 5 + 4 + 1 + 1


@pytest.mark.parametrize("pooling", ["NONE", "SUM", "MEAN"])
def test_debug_initializer_known_answer(pooling):
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    m = _make([8, 8, 8], pooling, init_mode=IM.DEBUG)
    idx = torch.tensor(INDICES, dtype=torch.int64, device=DEV) + 100000 * 3  # key % 100000 is what counts
    off = torch.tensor(OFFSETS, dtype=torch.int64, device=DEV)
    out = m(idx, off).detach().cpu().numpy()
    k = np.array(INDICES, np.float32)
    if pooling == "NONE":
        assert (out == k[:, None]).all()
    else:
        for i in range(8):
            f, b = divmod(i, 2)
            bag = k[OFFSETS[i]:OFFSETS[i + 1]]
            exp = bag.sum() if pooling == "SUM" or bag.size == 0 else bag.sum() / bag.size
            assert np.allclose(out[b, f * 8:(f + 1) * 8], exp)


def _dense_twin_step(weights, opt_objs, idx, off, fmap, dims, pooling, B):
    """dense reference forward with torch ops on the twin weights; returns output tensor"""
    outs = []
    F = len(fmap)
    if pooling == "NONE":
        rows = []
        for i in range(F * B):
            f = i // B
            rows.append(weights[fmap[f]][idx[off[i]:off[i + 1]]])
        return torch.cat(rows, 0)
    per_b = [[None] * F for _ in range(B)]
    for i in range(F * B):
        f, b = divmod(i, B)
        r = weights[fmap[f]][idx[off[i]:off[i + 1]]]
        s = r.sum(0)
        if pooling == "MEAN" and r.shape[0] > 0:
            s = s / r.shape[0]
        per_b[b][f] = s
    return torch.stack([torch.cat(per_b[b]) for b in range(B)])


@pytest.mark.parametrize("pooling", ["NONE", "SUM", "MEAN"])
@pytest.mark.parametrize("opt,params", [("SGD", dict(learning_rate=0.3)),
                                        ("ADAM", dict(learning_rate=0.03, eps=1e-8, beta1=0.9, beta2=0.999)),
                                        ("EXACT_ADAGRAD", dict(learning_rate=0.3, eps=1e-10))])
def test_backward_twin_10_iterations(pooling, opt, params):
    """Train 10 iterations on random batches against a dense twin (torch.nn parameters + torch.optim
    with sparse-equivalent semantics: only touched rows are updated)."""
    dims = [8, 8, 8]
    fmap = [0, 0, 1, 2]
    m = _make(dims, pooling, opt, cap=4096, **params)
    rng = np.random.default_rng(0)
    nkeys = 60
    B, F = 4, 4
    # materialise the table rows by touching all keys once, then copy them into the twin
    allk = torch.arange(nkeys, dtype=torch.int64, device=DEV)
    twin = []
    for t in range(3):
        off0 = torch.zeros(F * nkeys + 1, dtype=torch.int64, device=DEV)
        # feature f of table t holds all keys, batch = nkeys (one key per bag); other features empty
        feats = [i for i, tt in enumerate(fmap) if tt == t]
        lens = torch.zeros(F, nkeys, dtype=torch.int64, device=DEV)
        lens[feats[0]] = 1
        off0[1:] = torch.cumsum(lens.flatten(), 0)
        with torch.no_grad():
            m.train()
            m._forward_impl(allk, off0, train=True)
        found, rows = m.lookup_rows(allk, t)
        assert found.all().item()
        twin.append(rows[:, :dims[t]].clone().double().requires_grad_(True))
    if opt == "SGD":
        tops = [torch.optim.SGD([w], lr=params["learning_rate"]) for w in twin]
    elif opt == "ADAM":
        tops = None  # manual sparse Adam below (torch.optim.Adam would decay untouched rows' moments)
        mom = [torch.zeros_like(w) for w in twin]
        var = [torch.zeros_like(w) for w in twin]
    else:
        acc = [torch.zeros_like(w) for w in twin]
    for it in range(1, 11):
        lens = rng.integers(0, 4, size=F * B)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        idx = rng.integers(0, nkeys, size=int(off[-1])).astype(np.int64)
        ti, to = torch.from_numpy(idx).to(DEV), torch.from_numpy(off).to(DEV)
        out = m(ti, to)
        ref = _dense_twin_step(twin, None, ti, off, fmap, dims, pooling, B)
        torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-5)
        g = torch.randn_like(out)
        out.backward(g)
        for w in twin:
            w.grad = None
        ref.backward(g.double())
        with torch.no_grad():
            for t, w in enumerate(twin):
                gr = w.grad
                touched = (gr != 0).any(1) if gr is not None else None
                if gr is None:
                    continue
                # rows that occurred in the batch (even with zero grad) are the ones updated; use occurrence
                feats = [i for i, tt in enumerate(fmap) if tt == t]
                occ = torch.zeros(nkeys, dtype=torch.bool, device=DEV)
                for f in feats:
                    for b in range(B):
                        i = f * B + b
                        occ[ti[off[i]:off[i + 1]]] = True
                if opt == "SGD":
                    w[occ] -= params["learning_rate"] * gr[occ]
                elif opt == "ADAM":
                    b1, b2 = params["beta1"], params["beta2"]
                    mom[t][occ] = b1 * mom[t][occ] + (1 - b1) * gr[occ]
                    var[t][occ] = b2 * var[t][occ] + (1 - b2) * gr[occ] ** 2
                    mh = mom[t][occ] / (1 - b1 ** it)
                    vh = var[t][occ] / (1 - b2 ** it)
                    w[occ] -= params["learning_rate"] * (mh / (vh.sqrt() + params["eps"]))
                else:
                    acc[t][occ] += gr[occ] ** 2
                    w[occ] -= params["learning_rate"] * gr[occ] / (acc[t][occ].sqrt() + params["eps"])
    m.eval()
    for t in range(3):
        found, rows = m.lookup_rows(allk, t)
        torch.testing.assert_close(rows[:, :dims[t]].double(), twin[t].detach(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("strategy", ["STEP", "CUSTOMIZED", "LFU"])
def test_score_strategies_drive_eviction(strategy):
    """Small table (one bucket per table): later / more frequent keys survive, as the score policy says."""
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    m = B2(table_options=[TO(dim=8, max_capacity=128, embedding_dtype=torch.float32, bucket_capacity=128,
                             score_strategy=getattr(SS, strategy),
                             initializer_args=IA(mode=IM.DEBUG))],
           feature_table_map=[0], pooling_mode=PM.NONE, device=torch.device(DEV))
    off = lambda n: torch.arange(n + 1, dtype=torch.int64, device=DEV)
    old = torch.arange(1000, 1128, dtype=torch.int64, device=DEV)
    if strategy == "CUSTOMIZED":
        m.set_score(1)
    if strategy == "LFU":
        hot = old[:64]
        m(torch.cat([old, hot, hot]), off(256))  # hot keys counted 3x
    else:
        m(old, off(128))
    assert int(m.size().item()) == 128
    new = torch.arange(5000, 5032, dtype=torch.int64, device=DEV)
    if strategy == "CUSTOMIZED":
        m.set_score(2)
    out = m(new, off(32))
    assert (out[:, 0].detach().cpu().numpy() == np.arange(5000, 5032)).all()
    m.eval()
    back = m(old, off(128))
    alive = (back[:, 0] != 0).cpu().numpy()
    assert alive.sum() == 128 - 32
    if strategy == "LFU":
        assert alive[:64].all()  # the frequently used keys were kept


def test_bf16_output_matches_fp32_within_1e3():
    m = _make([128, 128, 128], "SUM", out_dtype=torch.bfloat16)
    m2 = _make([128, 128, 128], "SUM", out_dtype=torch.float32)
    rng = np.random.default_rng(1)
    B, F = 64, 4
    lens = rng.integers(1, 11, size=F * B)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(DEV)
    idx = torch.from_numpy(rng.integers(0, 500, size=int(lens.sum())).astype(np.int64)).to(DEV)
    a = m(idx, off).float()
    b = m2(idx, off)
    # same seed + counter based initialiser => identical rows in both modules
    assert torch.allclose(a, b, rtol=8e-3, atol=1e-3)  # bf16 has 8 bits of mantissa: half-ulp = 3.9e-3
    assert torch.allclose(a, b.bfloat16().float())       # i.e. exactly the rounded fp32 result


# ---------------------------------------------------------------------------------------- storage tiers
@pytest.mark.parametrize("mode", ["host", "hybrid", "cache"])
@pytest.mark.parametrize("pooling", ["SUM", "NONE"])
@pytest.mark.parametrize("optimizer", ["SGD", "ADAM"])
def test_storage_tiers_are_transparent(mode, pooling, optimizer):
    """Host-only storage and the hybrid HBM + host tiers must be invisible in the results: same outputs and same rows
    as the HBM-only module on the same key stream, including while the (tiny) HBM tier keeps evicting into the host
    tier (reference: test_batched_dynamic_embedding_tables_v2.py hybrid / host-only parametrisations :862-1318)."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    dims, F, B = [8, 8], 2, 24
    opt_t = EmbOptimType.SGD if optimizer == "SGD" else EmbOptimType.ADAM
    pm = DynamicEmbPoolingMode.SUM if pooling == "SUM" else DynamicEmbPoolingMode.NONE

    def make(storage_mode, local_hbm=0):
        caching = storage_mode == "cache"
        if caching:
            storage_mode = None   # selected through the options, as the reference does (caching=True + an HBM budget)
        opts = [DynamicEmbTableOptions(dim=d, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                       score_strategy=DynamicEmbScoreStrategy.STEP, local_hbm_for_values=local_hbm, caching=caching,
                                       initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG))
                for d in dims]
        m = BatchedDynamicEmbeddingTablesV2(opts, pooling_mode=pm, output_dtype=torch.float32, optimizer=opt_t,
                                            learning_rate=0.05, device=torch.device("cuda", 0), storage_mode=storage_mode)
        m.train()
        return m

    ref = make("hbm")
    # hybrid: room for 128 rows per table in HBM (one bucket), everything else spills to the host tier
    row_bytes = 4 * (dims[0] * (3 if optimizer == "ADAM" else 1))
    dut = make(mode, local_hbm=2 * 128 * row_bytes if mode in ("hybrid", "cache") else 0)
    assert dut.storage_mode == ("hybrid" if mode == "cache" else mode) and dut._promote == (mode == "cache")
    seen = set()
    for step in range(8):
        lens = rng.integers(0, 5, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        keys_np = rng.integers(0, 600, off[-1]).astype(np.int64)
        seen.update(keys_np.tolist())
        keys = torch.from_numpy(keys_np).cuda()
        off_t = torch.from_numpy(off).cuda()
        o_ref, st = ref._forward_impl(keys, off_t, train=True)
        o_dut, st2 = dut._forward_impl(keys, off_t, train=True)
        torch.testing.assert_close(o_dut, o_ref, rtol=1e-6, atol=1e-6)
        g = torch.randn_like(o_ref)
        ref._backward_impl(st, g)
        dut._backward_impl(st2, g)
        if step % 3 == 2:   # eval lookups in between (unknown keys -> zeros, no inserts)
            ek = torch.from_numpy(rng.integers(0, 1200, 40).astype(np.int64)).cuda()
            eo = torch.arange(0, 41, dtype=torch.int64, device="cuda")[: (40 // F) * F + 1]
            e1, _ = ref._forward_impl(ek[: eo.numel() - 1], eo, train=False)
            e2, _ = dut._forward_impl(ek[: eo.numel() - 1], eo, train=False)
            torch.testing.assert_close(e2, e1, rtol=1e-6, atol=1e-6)
    probe = torch.arange(0, 600, device="cuda", dtype=torch.int64)
    for t in range(len(dims)):
        f1, r1 = ref.lookup_rows(probe, t)
        f2, r2 = dut.lookup_rows(probe, t)
        assert torch.equal(f1, f2)
        torch.testing.assert_close(r2, r1, rtol=1e-5, atol=1e-6)
    if mode in ("hybrid", "cache"):
        assert int(dut.table_host.size()) > 0, "the HBM tier never spilled: the test does not exercise eviction"
        assert int(dut.size()) == int(ref.size())
    if mode == "cache":
        # a key that sits in the host tier comes back to the HBM tier when it is used again
        from dynamicemb.scored_hashtable import ScoreArg
        import dynamicemb_extensions as e

        allk = probe[dut.lookup_rows(probe, 0)[0]]
        tid0 = torch.zeros_like(allk)
        _, in_host, _ = dut.table_host.lookup(allk, tid0, ScoreArg("score", None, e.ScorePolicy.CONST))
        victim = allk[in_host][:4].contiguous()
        assert victim.numel() > 0
        offv = torch.arange(0, F * B + 1, device="cuda", dtype=torch.int64).clamp(max=victim.numel())
        dut._forward_impl(victim, offv, train=True)
        _, now_hbm, _ = dut.table.lookup(victim, torch.zeros_like(victim), ScoreArg("score", None, e.ScorePolicy.CONST))
        _, still_host, _ = dut.table_host.lookup(victim, torch.zeros_like(victim), ScoreArg("score", None, e.ScorePolicy.CONST))
        assert bool(now_hbm.all()) and not bool(still_host.any())


@pytest.mark.parametrize("optimizer", ["SGD", "ADAM", "ROWWISE"])
@pytest.mark.parametrize("strategy", ["STEP", "TIMESTAMP"])
def test_dump_load_wire_format_roundtrip(tmp_path, optimizer, strategy):
    """dump() writes the reference's checkpoint files (raw little-endian keys i64 / values f32 / scores i64 / opt f32 +
    {table}_opt_args.json, batched_dynamicemb_tables.py:73-92,1262-1409); load() into a fresh module restores every
    key, embedding, optimizer state and score-derived behaviour"""
    import json

    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    torch.manual_seed(1)
    rng = np.random.default_rng(1)
    dims, F, B = [8, 16], 2, 16
    opt_t = {"SGD": EmbOptimType.SGD, "ADAM": EmbOptimType.ADAM, "ROWWISE": EmbOptimType.EXACT_ROWWISE_ADAGRAD}[optimizer]
    strat = DynamicEmbScoreStrategy.STEP if strategy == "STEP" else DynamicEmbScoreStrategy.TIMESTAMP

    def make():
        opts = [DynamicEmbTableOptions(dim=d, max_capacity=2048, index_type=torch.int64, embedding_dtype=torch.float32,
                                       score_strategy=strat,
                                       initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.NORMAL))
                for d in dims]
        m = BatchedDynamicEmbeddingTablesV2(opts, table_names=["user", "item"], pooling_mode=DynamicEmbPoolingMode.SUM,
                                            output_dtype=torch.float32, optimizer=opt_t, learning_rate=0.05,
                                            device=torch.device("cuda", 0))
        m.train()
        return m

    src = make()
    for step in range(4):
        lens = rng.integers(0, 5, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        keys = torch.from_numpy(rng.integers(0, 500, off[-1]).astype(np.int64)).cuda()
        o, st = src._forward_impl(keys, torch.from_numpy(off).cuda(), train=True)
        src._backward_impl(st, torch.randn_like(o))
    src.dump(str(tmp_path), optim=True)
    # the files are what the reference's reader expects
    for t, name in enumerate(["user", "item"]):
        n = int(src.table.size(t))
        stem = lambda item: tmp_path / f"{name}_emb_{item}.rank_0.world_size_1"  # noqa: E731
        assert stem("keys").stat().st_size == 8 * n and stem("values").stat().st_size == 4 * dims[t] * n
        assert stem("scores").stat().st_size == 8 * n
        cs = {"SGD": 0, "ADAM": 2 * dims[t], "ROWWISE": 1}[optimizer]
        assert stem("opt_values").stat().st_size == 4 * cs * n
        meta = json.load(open(tmp_path / f"{name}_opt_args.json"))
        assert meta["opt_type"] == {"SGD": "sgd", "ADAM": "adam", "ROWWISE": "exact_row_wise_adagrad"}[optimizer]
        assert meta["dist_type"] == "roundrobin" and "evict_strategy" in meta
        kk = np.fromfile(stem("keys"), dtype=np.int64)
        assert len(set(kk.tolist())) == n
    dst = make()
    dst.load(str(tmp_path), optim=True)
    probe = torch.arange(0, 500, device="cuda", dtype=torch.int64)
    for t in range(2):
        f1, r1 = src.lookup_rows(probe, t)
        f2, r2 = dst.lookup_rows(probe, t)
        assert torch.equal(f1, f2) and int(f1.sum()) > 0
        D = dims[t]
        cs = {"SGD": 0, "ADAM": 2 * D, "ROWWISE": 1}[optimizer]
        assert torch.equal(r1[:, :D + cs], r2[:, :D + cs])
    # and training continues identically from the restored state (same optimizer step counter for Adam)
    lens = rng.integers(1, 4, F * B)
    off = np.zeros(F * B + 1, np.int64)
    off[1:] = np.cumsum(lens)
    keys = torch.from_numpy(rng.integers(0, 500, off[-1]).astype(np.int64)).cuda()
    off_t = torch.from_numpy(off).cuda()
    o1, s1 = src._forward_impl(keys, off_t, train=True)
    o2, s2 = dst._forward_impl(keys, off_t, train=True)
    known = src.lookup_rows(keys, 0)[0]   # rows that existed before this step are identical; new ones are random-initialised
    g = torch.randn_like(o1)
    src._backward_impl(s1, g)
    dst._backward_impl(s2, g)
    f1, r1 = src.lookup_rows(probe, 0)
    f2, r2 = dst.lookup_rows(probe, 0)
    both = f1 & f2
    # keys that were in the checkpoint received the same update on both sides
    ck = torch.from_numpy(np.fromfile(tmp_path / "user_emb_keys.rank_0.world_size_1", dtype=np.int64)).cuda()
    sel = torch.isin(probe, ck) & both
    torch.testing.assert_close(r1[sel][:, :dims[0]], r2[sel][:, :dims[0]], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("api", ["prefetch", "prefetch_async"])
@pytest.mark.parametrize("strategy,pooling,cap", [("STEP", "SUM", 1 << 20), ("TIMESTAMP", "SUM", 1 << 20), ("STEP", "NONE", 1 << 20),
                                                   ("STEP", "SUM", 1 << 17)])
def test_prefetch_one_batch_ahead_on_the_partitioned_index_path_matches_plain_training(api, strategy, pooling, cap):
    """Round 6: batches of the partitioned index path (>= 64 K keys, one table) are prefetched on it -- stage 1 (probe + partition
    kernel) in prefetch(), the gather alone in forward(); no ref-counter pin: evictions spare every slot that scores at least the
    oldest step in flight.  Same order as the test above (prefetch(k + 1) before backward(k)): outputs and final rows equal plain
    forward / backward; `prefetch_async` is the form with library-owned stream ordering.  cap = 1 << 17: the table is smaller than
    the key space, buckets fill up and the partition kernel evicts while batches are in flight."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    pm = DynamicEmbPoolingMode.SUM if pooling == "SUM" else DynamicEmbPoolingMode.NONE
    hi = 150_000 if cap == 1 << 20 else 400_000

    def make():
        opt = DynamicEmbTableOptions(dim=16, max_capacity=cap, index_type=torch.int64, embedding_dtype=torch.float32,
                                     score_strategy=getattr(DynamicEmbScoreStrategy, strategy),
                                     initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG))
        m = BatchedDynamicEmbeddingTablesV2([opt], pooling_mode=pm, output_dtype=torch.float32, optimizer=EmbOptimType.SGD,
                                            learning_rate=0.05, device=torch.device("cuda", 0))
        m.train()
        return m

    batches = []
    for _ in range(5):
        lens = rng.integers(1, 8, 20_000)
        off = np.zeros(lens.size + 1, np.int64)
        off[1:] = np.cumsum(lens)
        keys = (rng.zipf(1.3, off[-1]) % hi).astype(np.int64)
        if pooling == "NONE":
            off = np.arange(off[-1] + 1, dtype=np.int64)
        batches.append((torch.from_numpy(keys).cuda(), torch.from_numpy(off).cuda()))
    assert all(k.numel() >= 65_536 for k, _ in batches)
    plain, piped = make(), make()
    grads, outs_plain = [], []
    for k, o in batches:
        out, st = plain._forward_impl(k, o, train=True)
        g = torch.randn_like(out)
        grads.append(g)
        outs_plain.append(out)
        plain._backward_impl(st, g)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def pf(i):
        if api == "prefetch_async":
            piped.prefetch_async(*batches[i])
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                piped.prefetch(*batches[i], forward_stream=main)

    pf(0)
    for i, (k, o) in enumerate(batches):
        assert getattr(piped._prefetch_states[0], "staged", False), "the batch was not prefetched on the partitioned path"
        out, st = piped._forward_impl(k, o, train=True)       # consumes the prefetched state: the gather alone
        if i + 1 < len(batches):
            pf(i + 1)                                          # runs ahead of this batch's backward
        # (two runs of the partitioned path sum a row's gradients in the order their records were reserved: fp32 rows differ
        #  by an ulp or two between ANY two runs, the rows here are as large as 1e5)
        torch.testing.assert_close(out, outs_plain[i], rtol=1e-5, atol=1e-4)
        piped._backward_impl(st, grads[i])
        main.wait_stream(side)
    torch.cuda.synchronize()
    assert piped._pf_c_used and not piped._prefetch_states and len(piped._inflight) == 0
    assert int(piped.size()) == int(plain.size())
    probe = torch.arange(0, hi, device="cuda", dtype=torch.int64)
    f1, r1 = plain.lookup_rows(probe, 0)
    f2, r2 = piped.lookup_rows(probe, 0)
    if cap == 1 << 20:
        assert torch.equal(f1, f2)
        torch.testing.assert_close(r2, r1, rtol=1e-5, atol=1e-4)
    else:
        # evictions: which of two equal-score victims goes may differ between the schedules (the prefetch spares the rows of
        # the step in flight); every key both tables hold has the same row
        both = f1 & f2
        assert int(both.sum()) > 0.8 * min(int(f1.sum()), int(f2.sum()))
        torch.testing.assert_close(r2[both], r1[both], rtol=1e-5, atol=1e-4)
    assert int(piped.table._ref_counter.abs().sum()) == 0


@pytest.mark.parametrize("pooling", ["SUM", "NONE"])
def test_prefetch_one_batch_ahead_matches_plain_training(pooling):
    """prefetch(batch i+1) on a side stream before backward(batch i) (PrefetchTrainPipelineSparseDist's order,
    examples/commons/pipeline/train_pipeline.py:533-692): outputs and final rows equal those of plain
    forward / backward, and every pin is released at the end"""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    rng = np.random.default_rng(2)
    torch.manual_seed(2)
    dims, F, B = [8, 8], 2, 20
    pm = DynamicEmbPoolingMode.SUM if pooling == "SUM" else DynamicEmbPoolingMode.NONE

    def make():
        opts = [DynamicEmbTableOptions(dim=d, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                       score_strategy=DynamicEmbScoreStrategy.STEP,
                                       initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG))
                for d in dims]
        m = BatchedDynamicEmbeddingTablesV2(opts, pooling_mode=pm, output_dtype=torch.float32, optimizer=EmbOptimType.ADAM,
                                            learning_rate=0.05, device=torch.device("cuda", 0))
        m.train()
        return m

    batches = []
    for _ in range(6):
        lens = rng.integers(0, 5, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        batches.append((torch.from_numpy(rng.integers(0, 300, off[-1]).astype(np.int64)).cuda(), torch.from_numpy(off).cuda()))
    plain, piped = make(), make()
    grads, outs_plain = [], []
    for k, o in batches:
        out, st = plain._forward_impl(k, o, train=True)
        g = torch.randn_like(out)
        grads.append(g)
        outs_plain.append(out)
        plain._backward_impl(st, g)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    with torch.cuda.stream(side):
        piped.prefetch(*batches[0], forward_stream=main)
    for i, (k, o) in enumerate(batches):
        out, st = piped._forward_impl(k, o, train=True)       # consumes the prefetched state
        if i + 1 < len(batches):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                piped.prefetch(*batches[i + 1], forward_stream=main)   # runs ahead of this batch's backward
        torch.testing.assert_close(out, outs_plain[i], rtol=1e-6, atol=1e-6)
        piped._backward_impl(st, grads[i])
        main.wait_stream(side)
    torch.cuda.synchronize()
    probe = torch.arange(0, 300, device="cuda", dtype=torch.int64)
    for t in range(2):
        f1, r1 = plain.lookup_rows(probe, t)
        f2, r2 = piped.lookup_rows(probe, t)
        assert torch.equal(f1, f2)
        torch.testing.assert_close(r2, r1, rtol=1e-5, atol=1e-6)
    assert int(piped.table._ref_counter.abs().sum()) == 0, "a prefetched row stayed pinned"
    assert not piped._prefetch_states


# ---------------------------------------------------------------------------------------- BASELINE configs at full size
def _zipf_batch(rows, alpha, B, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    w = torch.arange(1, rows + 1, device="cuda", dtype=torch.float64).pow_(-alpha)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    lens = torch.randint(1, 11, (B,), device="cuda", generator=g)
    off = torch.zeros(B + 1, dtype=torch.int64, device="cuda")
    off[1:] = torch.cumsum(lens, 0)
    u = torch.rand(int(off[-1]), device="cuda", dtype=torch.float64, generator=g)
    perm = torch.randperm(rows, device="cuda", generator=g)
    return perm[torch.searchsorted(cdf, u).clamp_(max=rows - 1)].contiguous(), off


@pytest.mark.parametrize("B", [65536, 1048576])
def test_c2_full_size_properties(B):
    """BASELINE configs[1] at full size (10 M x 128 fp32, 65 536 bags, Zipf 0.99) and at the 16x batch SURVEY 8(d) asks to
    report (1 M bags, 5.8 M keys, the hottest row has > 300 K occurrences): no oracle can run this in seconds, so
    the checks are size-independent properties -- dedup round trip, every key found after insertion, the checksum of the
    pooled output equals the count-weighted checksum of the looked-up rows, and one SGD step with an all-ones gradient
    moves every row by exactly lr x (its number of occurrences)."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)
    import dynamicemb_extensions as e

    rows, D, lr = 10_000_000, 128, 0.25
    opt = DynamicEmbTableOptions(dim=D, max_capacity=rows, index_type=torch.int64, embedding_dtype=torch.float32,
                                 score_strategy=DynamicEmbScoreStrategy.TIMESTAMP,
                                 initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM, lower=-1, upper=1))
    m = BatchedDynamicEmbeddingTablesV2([opt], pooling_mode=DynamicEmbPoolingMode.SUM, output_dtype=torch.float32,
                                        optimizer=EmbOptimType.SGD, learning_rate=lr, device=torch.device("cuda", 0))
    m.train()
    keys, off = _zipf_batch(rows, 0.99, B, 5)
    nt = keys.numel()
    out, st = m._forward_impl(keys, off, train=True)
    # dedup round trip + counts
    nu = int(st.uoff[-1])
    rng_t = e.get_table_range(off, m.feature_offsets)
    uk, rev, uoff, cnt, rank = e.segmented_unique_csr(keys, rng_t, 1)
    assert int(uoff[-1]) == nu and torch.equal(uk[:nu][rev], keys)
    # the module's reverse indices describe the same partition of the batch (the fused forward numbers the uniques in
    # representative order, the per-op kernel in first-occurrence order; the reference's order is schedule dependent)
    relabel = torch.full((nu,), -1, dtype=torch.int64, device="cuda")
    relabel[rev] = st.rev
    assert torch.equal(relabel[rev], st.rev) and torch.equal(torch.sort(relabel).values, torch.arange(nu, device="cuda"))
    assert int(cnt[:nu].sum()) == nt and torch.equal(torch.bincount(rev, minlength=nu).to(torch.int32), cnt[:nu])
    assert nu == torch.unique(keys).numel()
    # every key of the batch is in the table, once
    found, rows0 = m.lookup_rows(uk[:nu].contiguous(), 0)
    assert bool(found.all()) and int(m.size()) == nu
    # checksum of checksums: sum of all pooled outputs == sum_u cnt[u] * row[u]   (fp64 on both sides)
    lhs = out.double().sum(0)
    rhs = (rows0[:, :D].double() * cnt[:nu].double()[:, None]).sum(0)
    # (every bag is one fp32 sum: the checksum carries ~sqrt(B) of their rounding errors)
    torch.testing.assert_close(lhs, rhs, rtol=1e-6, atol=1e-3 * (B / 65536) ** 0.5 * 4)
    # and bag by bag against an fp64 torch reference built from the looked-up rows
    bag = torch.repeat_interleave(torch.arange(B, device="cuda"), off[1:] - off[:-1])
    ref = torch.zeros(B, D, dtype=torch.float64, device="cuda").index_add_(0, bag, rows0[:, :D][rev].double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-6, atol=2e-6)
    del ref, bag
    # one SGD step, gradient of ones: w' = w - lr * bf16(occurrences) -- the reduced gradient is rounded to the gradient
    # dtype once before the update, as the reference's reduce_grads returns it (dynamic_emb_op.cu:159-285)
    m._backward_impl(st, torch.ones(B, D, device="cuda", dtype=torch.bfloat16))
    _, rows1 = m.lookup_rows(uk[:nu].contiguous(), 0)
    g_sum = cnt[:nu].float().bfloat16().float()
    torch.testing.assert_close(rows1[:, :D], rows0[:, :D] - lr * g_sum[:, None], rtol=0, atol=2e-6 * float(cnt[:nu].max()))
    # idempotence of the steady state: a second forward of the same batch inserts nothing
    out2, _ = m._forward_impl(keys, off, train=True)
    assert int(m.size()) == nu
    torch.testing.assert_close(out2.double().sum(0), (rows1[:, :D].double() * cnt[:nu].double()[:, None]).sum(0), rtol=1e-6, atol=1e-3)


def test_c1_matches_torch_embedding_bag():
    """BASELINE configs[0]: 1 table x 100 K rows x 32-D, batch 512 -- the CPU path a TorchRec EmbeddingBagCollection runs
    is torch.nn.functional.embedding_bag(mode='sum'); load the same weights, compare the pooled output and one sparse SGD
    step (no reference test pins the CPU EBC output, so this is the pin)."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbPoolingMode, DynamicEmbScoreStrategy, DynamicEmbTableOptions,
                                              EmbOptimType)

    rows, D, B, lr = 100_000, 32, 512, 0.1
    torch.manual_seed(0)
    weight = torch.randn(rows, D)
    opt = DynamicEmbTableOptions(dim=D, max_capacity=2 * rows, index_type=torch.int64, embedding_dtype=torch.float32,
                                 score_strategy=DynamicEmbScoreStrategy.STEP)
    m = BatchedDynamicEmbeddingTablesV2([opt], pooling_mode=DynamicEmbPoolingMode.SUM, output_dtype=torch.float32,
                                        optimizer=EmbOptimType.SGD, learning_rate=lr, device=torch.device("cuda", 0))
    m.train()
    m._insert_rows(0, torch.arange(rows, device="cuda"), weight.cuda(), torch.ones(rows, dtype=torch.int64, device="cuda"))
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(0, 20, (B,), generator=g)
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    keys = torch.randint(0, rows, (int(off[-1]),), generator=g)
    w_cpu = weight.clone().requires_grad_()
    ref = torch.nn.functional.embedding_bag(keys, w_cpu, off, mode="sum", include_last_offset=True, sparse=False)
    out, st = m._forward_impl(keys.cuda(), off.cuda(), train=True)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-6, atol=1e-5)
    grad = torch.randn(B, D)
    ref.backward(grad)
    m._backward_impl(st, grad.cuda())
    touched = torch.unique(keys)
    _, rows1 = m.lookup_rows(touched.cuda(), 0)
    torch.testing.assert_close(rows1.cpu(), (weight - lr * w_cpu.grad)[touched], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------ admission
@pytest.mark.parametrize("storage", ["hbm", "host", "hybrid", "cache"])
@pytest.mark.parametrize("prefetch", [False, True])
@pytest.mark.parametrize("threshold", [1, 3, 5])
@pytest.mark.parametrize("pooling", ["SUM", "NONE"])
def test_frequency_admission_against_dict_twin(threshold, pooling, prefetch, storage):
    """FrequencyAdmissionStrategy + KVCounter (reference test/unit_tests/test_embedding_admission.py: only keys whose
    accumulated frequency reached the threshold are stored).  Stronger than the reference's set invariant: a dict twin
    replays the admission rule step by step (batch frequency of every MISSING unique key is added to its counter; it is
    admitted -- and leaves the counter -- when the sum reaches the threshold; a rejected key is served a constant 0
    row and gets no update), so the stored key set, the counter population, every pooled output and every row after
    SGD must match.  prefetch: batch i + 1 walks the admission path (prefetch()) BEFORE batch i's backward, as the reference's
    prefetch pipeline does (batched_dynamicemb_function.py:559-696); the stored set, the outputs and the rows are the same.
    storage = "host": the table and its rows in pinned host memory (the same walk over the host link); "hybrid" / "cache"
    (round 4): a 128-row HBM tier over a host tier, with more keys than the HBM tier holds -- admitted keys take the two-tier
    insert walk (evictions spill down, "cache" also promotes), rejected ones are served from scratch rows."""
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    from dynamicemb.embedding_admission import FrequencyAdmissionStrategy, KVCounter

    D, F, B, steps, lr = 8, 2, 16, 6, 0.5
    strategy = FrequencyAdmissionStrategy(threshold=threshold, initializer_args=IA(mode=IM.CONSTANT, value=0.0))
    opts = [TO(dim=D, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
               initializer_args=IA(mode=IM.CONSTANT, value=0.25), score_strategy=SS.TIMESTAMP,
               admit_strategy=strategy, admission_counter=KVCounter(capacity=4096, bucket_capacity=128))]
    m = B2(table_options=opts, table_names=["t0"], feature_table_map=[0, 0], pooling_mode=getattr(PM, pooling),
           optimizer=OT.SGD, learning_rate=lr, output_dtype=torch.float32, device=torch.device(DEV), storage_mode=storage)
    m.train()
    assert m.storage_mode == ("hybrid" if storage == "cache" else storage)
    key_hi = 400 if storage in ("hybrid", "cache") else 40
    rng = np.random.default_rng(threshold * 7 + len(pooling))
    rows, counter = {}, {}
    batches = []
    for step in range(steps):
        lens = rng.integers(0, 4, F * B)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        batches.append((rng.integers(0, key_hi, int(off[-1])).astype(np.int64), off))
    dev = [(torch.from_numpy(k).to(DEV), torch.from_numpy(o).to(DEV)) for k, o in batches]
    if prefetch:
        m.prefetch(*dev[0])
    for step in range(steps):
        keys, off = batches[step]
        # ---- twin: admission on the unique keys of the batch
        uniq, cnt = np.unique(keys, return_counts=True)
        served = {}
        for k, c in zip(uniq.tolist(), cnt.tolist()):
            if k in rows:
                served[k] = True
                continue
            counter[k] = counter.get(k, 0) + c
            if counter[k] >= threshold:
                del counter[k]
                rows[k] = np.full(D, 0.25, np.float32)
                served[k] = True
            else:
                served[k] = False
        vec = lambda k: rows[k] if served[k] else np.zeros(D, np.float32)
        if pooling == "SUM":
            exp = np.zeros((B, F * D), np.float32)
            for f in range(F):
                for b in range(B):
                    for j in range(off[f * B + b], off[f * B + b + 1]):
                        exp[b, f * D:(f + 1) * D] += vec(int(keys[j]))
        else:
            exp = np.stack([vec(int(k)) for k in keys]) if keys.size else np.zeros((0, D), np.float32)
        out = m(*dev[step])
        if prefetch and step + 1 < steps:
            m.prefetch(*dev[step + 1])
        np.testing.assert_allclose(out.detach().cpu().numpy(), exp, rtol=1e-5, atol=1e-6)
        g = rng.standard_normal(exp.shape).astype(np.float32)
        out.backward(torch.from_numpy(g).to(DEV))
        # ---- twin: SGD on the stored rows only (the gradient of a rejected key is dropped)
        acc = {}
        if pooling == "SUM":
            for f in range(F):
                for b in range(B):
                    for j in range(off[f * B + b], off[f * B + b + 1]):
                        k = int(keys[j])
                        acc[k] = acc.get(k, 0) + g[b, f * D:(f + 1) * D]
        else:
            for j, k in enumerate(keys.tolist()):
                acc[k] = acc.get(k, 0) + g[j]
        for k, gk in acc.items():
            if served[k]:
                rows[k] = rows[k] - lr * gk
    got_k, got_v = m.export_keys_values("t0", torch.device("cpu"))
    got = {int(k): v.numpy() for k, v in zip(got_k, got_v)}
    assert set(got) == set(rows)
    for k, v in rows.items():
        np.testing.assert_allclose(got[k], v, rtol=1e-4, atol=1e-5)
    assert int(m._admission_counter.size()) == len(counter)
    assert m._tier_prefetched == 0 and int(m.table._ref_counter.sum()) == 0      # every pin was released by its backward


def test_admission_options_are_validated():
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    from dynamicemb.embedding_admission import FrequencyAdmissionStrategy

    with pytest.raises(ValueError):
        FrequencyAdmissionStrategy(threshold=-1)
    opts = [TO(dim=8, max_capacity=256, index_type=torch.int64, embedding_dtype=torch.float32,
               admit_strategy=FrequencyAdmissionStrategy(threshold=2))]
    with pytest.raises(ValueError):   # a strategy needs its counter
        B2(table_options=opts, pooling_mode=PM.SUM, device=torch.device(DEV))


@pytest.mark.parametrize("strategy", ["TIMESTAMP", "STEP"])
def test_incremental_dump_returns_rows_touched_since_the_threshold(strategy):
    """get_score() / incremental_dump() (reference batched_dynamicemb_tables.py:1166-1180,1432-1482): after a first
    batch, take the current score; only the keys touched by LATER batches (score >= threshold) come back, with their
    current rows; an unknown table name is skipped with a warning."""
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    D = 8
    opts = [TO(dim=D, max_capacity=1024, index_type=torch.int64, embedding_dtype=torch.float32,
               initializer_args=IA(mode=IM.UNIFORM, lower=-1.0, upper=1.0), score_strategy=getattr(SS, strategy))]
    m = B2(table_options=opts, table_names=["t0"], feature_table_map=[0], pooling_mode=PM.NONE, optimizer=OT.SGD,
           learning_rate=0.1, output_dtype=torch.float32, device=torch.device(DEV))
    m.train()

    def step(keys):
        k = torch.tensor(keys, dtype=torch.int64, device=DEV)
        off = torch.arange(len(keys) + 1, dtype=torch.int64, device=DEV)
        out = m(k, off)
        out.sum().backward()

    step([1, 2, 3, 4, 5])
    torch.cuda.synchronize()
    thr = m.get_score()["t0"] if strategy == "TIMESTAMP" else m.get_score()["t0"]
    step([4, 5, 6, 7])
    step([7, 8])
    with pytest.warns(UserWarning):
        got, now = m.incremental_dump({"t0": thr, "nope": 0})
    keys, vals = got["t0"]
    assert set(keys.tolist()) == {4, 5, 6, 7, 8} and "nope" not in got
    assert now["t0"] >= thr
    all_k, all_v = m.export_keys_values("t0", torch.device("cpu"))
    ref = {int(k): v for k, v in zip(all_k, all_v)}
    for k, v in zip(keys.tolist(), vals):
        torch.testing.assert_close(v.float(), ref[k])
    # threshold 0 returns everything
    everything, _ = m.incremental_dump({"t0": 0})
    assert set(everything["t0"][0].tolist()) == {1, 2, 3, 4, 5, 6, 7, 8}


# ------------------------------------------------------------------------------------------ early CSR
@pytest.mark.parametrize("pooling", ["SUM", "NONE"])
def test_early_csr_matches_late_grouping_and_survives_outstanding_steps(pooling):
    """The backward's key grouping forked onto the side stream by the forward (early CSR) must give the same table as
    grouping inside the backward -- also when MORE forwards are outstanding than the workspace ring holds (the extra
    ones fall back to late grouping), when a forward never gets its backward, and when backwards come in reverse."""
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    D, F, B = 16, 2, 64

    def make(early):
        opts = [TO(dim=D, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                   initializer_args=IA(mode=IM.UNIFORM, lower=-0.5, upper=0.5), score_strategy=SS.STEP)]
        m = B2(table_options=opts, table_names=["t0"], feature_table_map=[0, 0], pooling_mode=getattr(PM, pooling),
               optimizer=OT.ADAM, learning_rate=0.05, output_dtype=torch.float32, device=torch.device(DEV))
        m.train()
        m._early_csr = early
        return m

    rng = np.random.default_rng(11)
    batches = []
    for _ in range(7):
        lens = rng.integers(0, 6, F * B)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        keys = rng.zipf(1.3, int(off[-1])).astype(np.int64) % 500      # skewed: hot rows take the chunked paths
        n_out = B if pooling == "SUM" else int(off[-1])
        g = rng.standard_normal((n_out, F * D if pooling == "SUM" else D)).astype(np.float32)
        batches.append((torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV), torch.from_numpy(g).to(DEV)))
    tables = []
    for early in (True, False):
        m = make(early)
        outs = [m(k, o) for k, o, _ in batches[:6]]           # six outstanding forwards, ring of four
        with torch.enable_grad():
            m(batches[6][0], batches[6][1])                   # a forward that never gets a backward
        for out, (_, _, g) in reversed(list(zip(outs, batches[:6]))):
            out.backward(g)
        k, v = m.export_keys_values("t0", torch.device("cpu"))
        order = torch.argsort(k)
        tables.append((k[order], v[order]))
        assert not any(m._bwd_busy) or early is False or sum(m._bwd_busy) <= 1   # only the dropped step may linger until GC
    assert torch.equal(tables[0][0], tables[1][0])
    torch.testing.assert_close(tables[0][1], tables[1][1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,NK,cap,hi", [(128, 640, 8192, 3000), (8192, 70_000, 1 << 20, 300_000)],
                         ids=["per_slot_counters", "slot_range_partitions"])
def test_training_step_is_hipgraph_capturable(B, NK, cap, hi):
    """the step is a fixed launch sequence with every count on the device -- capture forward + backward of
    the module once in a HIP graph (static key / offset / gradient buffers), replay it on new batches, and compare the
    table with an eager twin fed the same batches.  The second size takes the partitioned index stage (its counters are
    cleared by the kernels themselves, so a replay starts clean)."""
    (B2, IA, IM, PM, SS, TO, OT) = _mods()
    D = 32
    from mi355_native import lib
    assert (lib().mi355_demb_forward_fused_partitions(NK, 1, cap // 128) > 0) == (NK >= 65536)

    def make():
        opts = [TO(dim=D, max_capacity=cap, index_type=torch.int64, embedding_dtype=torch.float32,
                   initializer_args=IA(mode=IM.UNIFORM, lower=-0.5, upper=0.5), score_strategy=SS.TIMESTAMP)]
        m = B2(table_options=opts, table_names=["t0"], feature_table_map=[0], pooling_mode=PM.SUM, optimizer=OT.SGD,
               learning_rate=0.1, output_dtype=torch.float32, device=torch.device(DEV))
        m.train()
        return m

    rng = np.random.default_rng(5)

    def batch():
        cuts = np.sort(rng.integers(0, NK + 1, B - 1))
        off = np.concatenate([[0], cuts, [NK]]).astype(np.int64)       # fixed number of keys, ragged bags
        keys = rng.integers(0, hi, NK).astype(np.int64)
        g = rng.standard_normal((B, D)).astype(np.float32)
        return torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV), torch.from_numpy(g).to(DEV)

    batches = [batch() for _ in range(4)]
    eager, graphed = make(), make()
    for k, o, g in batches:
        out, st = eager._forward_impl(k, o, train=True)
        eager._backward_impl(st, g)
    sk, so, sg = (t.clone() for t in batches[0])
    out0, st0 = graphed._forward_impl(sk, so, train=True)      # warm-up outside the capture (allocator, attributes)
    graphed._backward_impl(st0, sg)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out_g, st_g = graphed._forward_impl(sk, so, train=True)
            graphed._backward_impl(st_g, sg)
    for k, o, g in batches[1:]:
        sk.copy_(k); so.copy_(o); sg.copy_(g)
        graph.replay()
    torch.cuda.synchronize()
    ek, ev = eager.export_keys_values("t0", torch.device("cpu"))
    gk, gv = graphed.export_keys_values("t0", torch.device("cpu"))
    eo, go = torch.argsort(ek), torch.argsort(gk)
    assert torch.equal(ek[eo], gk[go])
    torch.testing.assert_close(ev[eo], gv[go], rtol=1e-5, atol=1e-6)


class _DictStore:
    """A store on the host, written for this test (the role of the reference's PyDictStorage parametrisation,
    test_batched_dynamic_embedding_tables_v2.py:862-1318): {(table id, key): row}, rows kept on the device."""

    def __new__(cls, options, optimizer):
        from dynamicemb.types import Storage

        class Impl(Storage):
            def __init__(self, options, optimizer):
                self.rows, self.scores = {}, {}
                self.emb_dim = max(o.dim for o in options)
                self.value_dim = max(o.dim + optimizer.get_state_dim(o.dim) for o in options)
                self.dtype = options[0].embedding_dtype
                self.finds = self.inserts = 0

            def size(self):
                return len(self.rows)

            def find(self, unique_keys, table_ids, copy_mode, lfu_accumulated_frequency=None):
                from dynamicemb.types import CopyMode

                self.finds += 1
                dev = unique_keys.device
                width = self.emb_dim if copy_mode == CopyMode.EMBEDDING else self.value_dim
                vals = torch.zeros(unique_keys.numel(), width, dtype=self.dtype, device=dev)
                ks, ts = unique_keys.cpu().tolist(), table_ids.cpu().tolist()
                found, miss = [], []
                for i, (k, t) in enumerate(zip(ks, ts)):
                    r = self.rows.get((t, k))
                    found.append(r is not None)
                    if r is None:
                        miss.append(i)
                    else:
                        vals[i] = r[:width]
                midx = torch.tensor(miss, dtype=torch.int64, device=dev)
                return (len(miss), unique_keys[midx], midx, table_ids[midx], None, torch.tensor(found, dtype=torch.bool, device=dev),
                        torch.zeros(unique_keys.numel(), dtype=torch.int64, device=dev), vals)

            def dump(self, table_id, meta_file_path, emb_key_path, embedding_file_path, score_file_path, opt_file_path, **kw):
                items = sorted((k, r) for (t, k), r in self.rows.items() if t == table_id)
                np.asarray([k for k, _ in items], np.int64).tofile(emb_key_path)
                rows = torch.stack([r for _, r in items]).float().cpu().numpy() if items else np.zeros((0, self.value_dim), np.float32)
                rows[:, :self.emb_dim].tofile(embedding_file_path)
                np.asarray([self.scores.get((table_id, k), 0) for k, _ in items], np.int64).tofile(score_file_path)
                if opt_file_path is not None:
                    rows[:, self.emb_dim:].tofile(opt_file_path)

            def load(self, table_id, meta_file_path, emb_key_path, embedding_file_path, score_file_path, opt_file_path, **kw):
                keys = np.fromfile(emb_key_path, np.int64)
                emb = np.fromfile(embedding_file_path, np.float32).reshape(len(keys), self.emb_dim)
                opt = (np.fromfile(opt_file_path, np.float32).reshape(len(keys), -1) if opt_file_path is not None
                       else np.zeros((len(keys), self.value_dim - self.emb_dim), np.float32))
                for k, e, o in zip(keys.tolist(), emb, opt):
                    self.rows[(table_id, k)] = torch.from_numpy(np.concatenate([e, o])).to("cuda")

            def insert(self, keys, table_ids, values, scores=None, preserve_existing=False):
                self.inserts += 1
                sc = scores.cpu().tolist() if scores is not None else None
                for i, (k, t) in enumerate(zip(keys.cpu().tolist(), table_ids.cpu().tolist())):
                    if preserve_existing and (t, k) in self.rows:
                        continue
                    self.rows[(t, k)] = values[i].clone()
                    if sc is not None:
                        self.scores[(t, k)] = sc[i]

        return Impl(options, optimizer)


@pytest.mark.parametrize("pooling", ["SUM", "MEAN", "NONE"])
@pytest.mark.parametrize("optimizer", ["SGD", "ADAM", "EXACT_ADAGRAD", "EXACT_ROWWISE_ADAGRAD"])
def test_external_storage_is_transparent(pooling, optimizer):
    """`DynamicEmbTableOptions.external_storage`: the rows live in a user-supplied `dynamicemb.types.Storage` (here a dict on
    the host); the module de-duplicates, finds, initialises + inserts the unknown keys, pools from the returned buffer, and in
    the backward reduces, steps the optimizer on the buffer and writes the rows back.  Same outputs and same rows as the
    HBM-only module on the same key stream (first-touch rows are keyed by the key in both), train and eval."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)
    from dynamicemb.external_storage import ExternalStorageTables

    rng = np.random.default_rng(5)
    dims, F, B = [8, 8], 2, 24
    pm = getattr(DynamicEmbPoolingMode, pooling)
    hp = dict(learning_rate=0.05)
    if optimizer == "ADAM":
        hp.update(beta1=0.8, beta2=0.9, eps=1e-6, weight_decay=0.01)
    elif optimizer != "SGD":
        hp.update(eps=1e-6, initial_accumulator_value=0.1)

    def make(store):
        opts = [DynamicEmbTableOptions(dim=d, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                       score_strategy=DynamicEmbScoreStrategy.STEP, external_storage=store,
                                       initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM, lower=-0.3, upper=0.3))
                for d in dims]
        m = BatchedDynamicEmbeddingTablesV2(opts, pooling_mode=pm, output_dtype=torch.float32, optimizer=getattr(EmbOptimType, optimizer),
                                            device=torch.device("cuda", 0), **hp)
        m.train()
        return m

    ref, dut = make(None), make(_DictStore)
    assert isinstance(dut, ExternalStorageTables) and isinstance(dut, BatchedDynamicEmbeddingTablesV2) and dut.storage_mode == "external"
    for step in range(6):
        lens = rng.integers(0, 5, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        keys = torch.from_numpy(rng.integers(0, 300, off[-1]).astype(np.int64)).cuda()
        off_t = torch.from_numpy(off).cuda()
        o_ref = ref(keys, off_t)
        o_dut = dut(keys, off_t)
        torch.testing.assert_close(o_dut, o_ref, rtol=1e-6, atol=1e-6)
        g = torch.randn_like(o_ref)
        o_ref.backward(g)
        o_dut.backward(g)
    assert dut.storage.finds == 6 and dut.storage.inserts >= 6 and dut.size() == int(ref.size())
    # every row the store holds equals the HBM module's row (embedding and optimizer state)
    probe = torch.arange(0, 300, device="cuda", dtype=torch.int64)
    for t in range(len(dims)):
        f1, r1 = ref.lookup_rows(probe, t)
        for k in probe[f1].tolist():
            torch.testing.assert_close(dut.storage.rows[(t, k)][: r1.size(1)], r1[k], rtol=1e-5, atol=1e-6)
        assert sum(1 for (tt, _k) in dut.storage.rows if tt == t) == int(f1.sum())
    # eval: known keys from the store, unknown keys -> the eval initializer (zeros), nothing inserted
    ref.eval(); dut.eval()
    ek = torch.from_numpy(rng.integers(0, 600, F * B).astype(np.int64)).cuda()
    eo = torch.arange(0, F * B + 1, dtype=torch.int64, device="cuda")
    n_before = dut.size()
    with torch.no_grad():
        torch.testing.assert_close(dut(ek, eo), ref(ek, eo), rtol=1e-6, atol=1e-6)
    assert dut.size() == n_before


@pytest.mark.parametrize("caching", [False, True], ids=["store_only", "hbm_cache_in_front"])
@pytest.mark.parametrize("pooling,optimizer,dims", [("SUM", "SGD", [8, 16, 32]), ("MEAN", "ADAM", [8, 16, 32]),
                                                    ("SUM", "EXACT_ROWWISE_ADAGRAD", [16, 8, 32]), ("NONE", "EXACT_ADAGRAD", [8, 8, 8])])
def test_external_storage_mixed_dims_and_hbm_cache(caching, pooling, optimizer, dims):
    """the two PS layouts of the reference's twin test (test_batched_dynamic_embedding_tables_v2.py:1414-1421, dims [8, 16, 32]):
    a store alone (HOST_PS) and a store behind an HBM cache (`caching=True`, CACHING_PS) whose 128 rows per table are far fewer
    than the keys in flight -- so rows are fetched, cached, evicted and written back all the time, and keys the cache refuses are
    trained in the spill buffer.  Value rows cross the `Storage` interface in the padded layout.  Same outputs as the HBM-only
    module at every step; after flush() the store holds exactly the HBM module's rows (embedding and optimizer state)."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)
    from dynamicemb.external_storage import ExternalStorageTables

    rng = np.random.default_rng(9)
    fmap = [0, 0, 1, 2]
    F, B = len(fmap), 40
    pm = getattr(DynamicEmbPoolingMode, pooling)
    hp = dict(learning_rate=0.05)
    if optimizer == "ADAM":
        hp.update(beta1=0.8, beta2=0.9, eps=1e-6, weight_decay=0.01)
    elif optimizer != "SGD":
        hp.update(eps=1e-6, initial_accumulator_value=0.1)

    def make(store):
        kw = dict(external_storage=store, caching=caching, local_hbm_for_values=1024) if store is not None else {}
        opts = [DynamicEmbTableOptions(dim=d, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                       score_strategy=DynamicEmbScoreStrategy.STEP,
                                       initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM, lower=-0.3, upper=0.3),
                                       **kw) for d in dims]
        m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=fmap, pooling_mode=pm, output_dtype=torch.float32,
                                            optimizer=getattr(EmbOptimType, optimizer), device=torch.device("cuda", 0), **hp)
        m.train()
        return m

    ref, dut = make(None), make(_DictStore)
    assert isinstance(dut, ExternalStorageTables) and (dut._cache is not None) == caching
    if caching:
        assert all(int(c) == dut._dynamicemb_options[0].bucket_capacity for c in dut._cache.table.per_table_capacity_)   # one bucket per table
    for step in range(8):
        lens = rng.integers(0, 6, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        keys = torch.from_numpy(rng.integers(0, 500, off[-1]).astype(np.int64)).cuda()
        off_t = torch.from_numpy(off).cuda()
        o_ref = ref(keys, off_t)
        o_dut = dut(keys, off_t)
        torch.testing.assert_close(o_dut, o_ref, rtol=1e-6, atol=1e-6, msg=f"step {step}")
        g = torch.randn_like(o_ref)
        o_ref.backward(g)
        o_dut.backward(g)
    if caching:
        assert dut.storage.inserts > 8, "nothing was ever evicted from the cache"
    dut.flush()
    assert dut.size() == int(ref.size())
    maxD = max(dims)
    probe = torch.arange(0, 500, device="cuda", dtype=torch.int64)
    for t, d in enumerate(dims):
        f1, r1 = ref.lookup_rows(probe, t)
        sdim = r1.size(1) - d
        for k in probe[f1].tolist():
            row = dut.storage.rows[(t, k)]
            torch.testing.assert_close(row[:d], r1[k][:d], rtol=1e-5, atol=1e-6)
            if sdim:
                torch.testing.assert_close(row[maxD:maxD + sdim], r1[k][d:], rtol=1e-5, atol=1e-6)
        assert sum(1 for (tt, _k) in dut.storage.rows if tt == t) == int(f1.sum())
    ref.eval(); dut.eval()
    ek = torch.from_numpy(rng.integers(0, 900, F * B).astype(np.int64)).cuda()
    eo = torch.arange(0, F * B + 1, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        torch.testing.assert_close(dut(ek, eo), ref(ek, eo), rtol=1e-6, atol=1e-6)


def test_external_storage_cache_keeps_the_rows_of_a_step_until_its_backward():
    """`caching=True` with TWO training forwards in flight (a shared module called twice / gradient accumulation): the second
    forward's inserts must not evict a cache row whose address the first step still holds (round-4 advisor finding: the stale
    row went back to the store and the first backward updated a slot that by then belonged to another key).  The steps' rows
    stay pinned until their backward; keys the cache then refuses are trained in the spill buffer.  Checked against the HBM-only
    module driven in the same order: same outputs, same rows in the store after flush()."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    rng = np.random.default_rng(31)
    dims, fmap = [8], [0]
    B = 100

    def make(store):
        kw = dict(external_storage=store, caching=True, local_hbm_for_values=1024) if store is not None else {}
        opts = [DynamicEmbTableOptions(dim=d, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                       score_strategy=DynamicEmbScoreStrategy.STEP,
                                       initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM, lower=-0.3, upper=0.3),
                                       **kw) for d in dims]
        m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=fmap, pooling_mode=DynamicEmbPoolingMode.SUM,
                                            output_dtype=torch.float32, optimizer=EmbOptimType.SGD, learning_rate=0.05,
                                            device=torch.device("cuda", 0))
        m.train()
        return m

    ref, dut = make(None), make(_DictStore)
    cap = int(dut._cache.table.per_table_capacity_[0])       # one 128-row bucket: two batches of ~90 distinct keys do not fit

    def batch(lo):
        keys = torch.from_numpy(rng.integers(lo, lo + 300, B).astype(np.int64)).cuda()
        return keys, torch.arange(0, B + 1, dtype=torch.int64, device="cuda")

    for it in range(6):
        (k1, o1), (k2, o2) = batch(0), batch(150)            # overlapping key ranges: shared rows, plus more rows than the cache holds
        assert int(torch.unique(torch.cat([k1, k2])).numel()) > cap
        outs = []
        for m in (ref, dut):
            a1 = m(k1, o1)
            a2 = m(k2, o2)                                   # second forward BEFORE the first backward
            outs.append((a1, a2))
        torch.testing.assert_close(outs[1][0], outs[0][0], rtol=1e-6, atol=1e-6, msg=f"iteration {it}, first forward")
        torch.testing.assert_close(outs[1][1], outs[0][1], rtol=1e-6, atol=1e-6, msg=f"iteration {it}, second forward")
        g1, g2 = torch.randn_like(outs[0][0]), torch.randn_like(outs[0][1])
        for a1, a2 in outs:
            a1.backward(g1)
            a2.backward(g2)
        torch.cuda.synchronize()
        assert int(dut._cache.table._ref_counter.abs().sum()) == 0, "pins left behind after both backwards"
    dut.flush()
    assert dut.size() == int(ref.size())
    probe = torch.arange(0, 450, device="cuda", dtype=torch.int64)
    f1, r1 = ref.lookup_rows(probe, 0)
    for k in probe[f1].tolist():
        torch.testing.assert_close(dut.storage.rows[(0, k)][:dims[0]], r1[k][:dims[0]], rtol=1e-5, atol=1e-6)


def test_external_storage_dump_and_load_go_through_the_store(tmp_path):
    """dump() / load() of a module over an external store hand the reference's per-table file names to Storage.dump / Storage.load
    (batched_dynamicemb_tables.py:73-92): a second module over a fresh store serves the same rows after load()."""
    import os

    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import DynamicEmbPoolingMode, DynamicEmbTableOptions, EmbOptimType

    def make():
        opts = [DynamicEmbTableOptions(dim=8, max_capacity=1024, index_type=torch.int64, embedding_dtype=torch.float32,
                                       external_storage=_DictStore) for _ in range(2)]
        return BatchedDynamicEmbeddingTablesV2(opts, table_names=["a", "b"], pooling_mode=DynamicEmbPoolingMode.NONE,
                                               optimizer=EmbOptimType.ADAM, learning_rate=0.1, device=torch.device("cuda", 0))

    m = make()
    m.train()
    keys = torch.arange(0, 40, device="cuda", dtype=torch.int64)
    off = torch.arange(0, 41, device="cuda", dtype=torch.int64)      # 2 features x 20 samples, one key each
    out = m(keys, off)
    out.backward(torch.ones_like(out))
    m.dump(str(tmp_path), optim=True)
    for name in ("a", "b"):
        for item in ("keys", "values", "scores", "opt_values"):
            assert os.path.exists(tmp_path / f"{name}_emb_{item}.rank_0.world_size_1"), (name, item)
        assert os.path.exists(tmp_path / f"{name}_opt_args.json")
    m2 = make()
    m2.load(str(tmp_path), optim=True)
    assert m2.size() == m.size() == 40
    m.eval(); m2.eval()
    with torch.no_grad():
        assert torch.equal(m2(keys, off), m(keys, off))
    for kk, r in m.storage.rows.items():
        assert torch.equal(m2.storage.rows[kk], r)


def test_external_storage_rejects_what_it_cannot_do():
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import DynamicEmbTableOptions

    mk = lambda **kw: [DynamicEmbTableOptions(dim=8, max_capacity=1024, index_type=torch.int64, embedding_dtype=torch.float32,  # noqa: E731
                                              external_storage=_DictStore, **kw)]
    m = BatchedDynamicEmbeddingTablesV2(mk(), device=torch.device("cuda", 0))
    with pytest.raises(NotImplementedError, match="prefetch"):
        m.prefetch(torch.zeros(1, dtype=torch.int64, device="cuda"), torch.tensor([0, 1], device="cuda"))


def test_the_prebound_plan_follows_the_hyper_parameters_it_froze():
    """the pre-bound training step (mi355_demb_plan_*) binds the initializer, the optimizer state's initial value and the score
    policy once; changing one of them later must reach the kernels (the plan is rebuilt at the next step), as it does on the
    general path"""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    def make():
        opt = DynamicEmbTableOptions(dim=8, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                     score_strategy=DynamicEmbScoreStrategy.STEP,
                                     initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.CONSTANT, value=1.5))
        m = BatchedDynamicEmbeddingTablesV2([opt], pooling_mode=DynamicEmbPoolingMode.NONE, output_dtype=torch.float32,
                                            optimizer=EmbOptimType.SGD, learning_rate=0.5, device=torch.device("cuda", 0))
        m.train()
        return m

    m = make()
    if not m._plan_ok:
        pytest.skip("the pre-bound step is not in use in this configuration")
    off = lambda n: torch.arange(n + 1, dtype=torch.int64, device="cuda")
    a = torch.arange(0, 50, dtype=torch.int64, device="cuda")
    out, st = m._forward_impl(a, off(50), train=True)
    assert getattr(st, "plan_step", False) and bool((out == 1.5).all())
    m._backward_impl(st, torch.zeros_like(out))
    key0 = m._plan_key
    m.initializer_args.value = -2.0                 # mutated in place: new keys must be initialised with it
    b = torch.arange(100, 150, dtype=torch.int64, device="cuda")
    out, st = m._forward_impl(b, off(50), train=True)
    assert m._plan_key != key0 and bool((out == -2.0).all())
    m._backward_impl(st, torch.zeros_like(out))
    out, st = m._forward_impl(a, off(50), train=True)    # keys of the first step keep their rows
    assert bool((out == 1.5).all())
    m._backward_impl(st, torch.zeros_like(out))
