"""f-rows of SURVEY 8 against an INDEPENDENT twin (oracle/dict_twin.py: python dicts + closed-form initialiser + the
restated optimizer maths) instead of against another mode of the product: storage tiers (HBM / host / hybrid / cache,
key_value_table.py:1522-2403), the prefetch pipeline (batched_dynamicemb_tables.py:1090-1137), the pre-communication dedup
of the sharded pooled path (shard/embedding.py:183-275) and the checkpoint wire format (batched_dynamicemb_tables.py
:73-92,1262-1409) read back by an independent numpy reader."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.dict_twin import DictEmbeddingTwin

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
_OPT = {"SGD": "sgd", "ADAM": "adam", "EXACT_ADAGRAD": "adagrad", "EXACT_ROWWISE_ADAGRAD": "rowwise_adagrad"}


def _module(dims, fmap, pooling, optimizer, lr, cap=4096, storage_mode=None, local_hbm=0, caching=False, **kw):
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    opts = [DynamicEmbTableOptions(dim=d, max_capacity=cap, index_type=torch.int64, embedding_dtype=torch.float32,
                                   score_strategy=DynamicEmbScoreStrategy.STEP, local_hbm_for_values=local_hbm, caching=caching,
                                   initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG)) for d in dims]
    m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=fmap, pooling_mode=getattr(DynamicEmbPoolingMode, pooling),
                                        output_dtype=torch.float32, optimizer=getattr(EmbOptimType, optimizer),
                                        learning_rate=lr, device=DEV, storage_mode=storage_mode, **kw)
    m.train()
    return m


def _batch(rng, F, B, hi, maxlen=5):
    lens = rng.integers(0, maxlen, F * B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    keys = rng.integers(0, hi, int(off[-1])).astype(np.int64)
    return keys, off


def _check_rows(m, twin, hi, rtol=2e-5, atol=2e-4):
    probe = np.arange(hi, dtype=np.int64)
    for t, d in enumerate(twin.dims):
        f_t, r_t = twin.rows(t, probe)
        f_m, r_m = m.lookup_rows(torch.from_numpy(probe).to(DEV), t)
        assert np.array_equal(f_m.cpu().numpy(), f_t)
        np.testing.assert_allclose(r_m[:, :d].cpu().numpy()[f_t], r_t[f_t], rtol=rtol, atol=atol)


@pytest.mark.parametrize("mode", ["hbm", "host", "hybrid", "cache"])
@pytest.mark.parametrize("pooling", ["SUM", "MEAN", "NONE"])
@pytest.mark.parametrize("optimizer", ["SGD", "ADAM", "EXACT_ROWWISE_ADAGRAD"])
def test_storage_tiers_against_the_dict_twin(mode, pooling, optimizer):
    dims, fmap, F, B, hi, lr = [8, 8], [0, 1, 1], 3, 16, 600, 0.05
    row_bytes = 4 * (dims[0] + {"SGD": 0, "ADAM": 16, "EXACT_ROWWISE_ADAGRAD": 4}[optimizer])
    kw = {}
    if mode in ("hybrid", "cache"):
        kw = dict(local_hbm=2 * 128 * row_bytes, caching=mode == "cache")     # one 128-row bucket per table in HBM
    m = _module(dims, fmap, pooling, optimizer, lr, storage_mode=None if mode in ("hybrid", "cache") else mode, **kw)
    assert m.storage_mode == ("hybrid" if mode == "cache" else mode)
    twin = DictEmbeddingTwin(dims, fmap, pooling, _OPT[optimizer], lr=lr)
    rng = np.random.default_rng(17)
    for step in range(8):
        keys, off = _batch(rng, F, B, hi)
        kt, ot = torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV)
        out, st = m._forward_impl(kt, ot, train=True)
        ref = twin.forward(keys, off, train=True)
        np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=1e-6, atol=1e-3)
        g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)      # (positive: well conditioned for Adam)
        m._backward_impl(st, torch.from_numpy(g).to(DEV))
        twin.backward(g)
        if step % 3 == 2:      # eval in between: unknown keys read zeros and are not inserted
            ek, eo = _batch(rng, F, 4, 2 * hi)
            e_out, _ = m._forward_impl(torch.from_numpy(ek).to(DEV), torch.from_numpy(eo).to(DEV), train=False)
            np.testing.assert_allclose(e_out.double().cpu().numpy(), twin.forward(ek, eo, train=False), rtol=1e-6, atol=1e-3)
    _check_rows(m, twin, hi)
    assert int(m.size()) == sum(len(t) for t in twin.tables)
    if mode in ("hybrid", "cache"):
        assert int(m.table_host.size()) > 0, "the HBM tier never spilled: eviction between the tiers was not exercised"


@pytest.mark.parametrize("pooling", ["SUM", "NONE"])
def test_prefetch_pipeline_against_the_dict_twin(pooling):
    """prefetch(batch i+1) is issued before backward(batch i), on a side stream, as the reference's prefetch pipeline does"""
    dims, fmap, F, B, hi, lr = [16], [0, 0], 2, 32, 400, 0.1
    m = _module(dims, fmap, pooling, "SGD", lr, cap=512, prefetch_pipeline=True)     # 4 buckets: prefetches meet full buckets
    twin = DictEmbeddingTwin(dims, fmap, pooling, "sgd", lr=lr)
    rng = np.random.default_rng(3)
    batches = [_batch(rng, F, B, hi) for _ in range(6)]
    dev = [(torch.from_numpy(k).to(DEV), torch.from_numpy(o).to(DEV)) for k, o in batches]
    side = torch.cuda.Stream()
    m.prefetch(*dev[0])
    for i, (keys, off) in enumerate(batches):
        out, st = m._forward_impl(*dev[i], train=True)
        if i + 1 < len(batches):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m.prefetch(*dev[i + 1])
        ref = twin.forward(keys, off, train=True)
        # a prefetched batch reads the rows as they are when its forward runs: after the previous backward
        np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=1e-6, atol=1e-3)
        g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)
        torch.cuda.current_stream().wait_stream(side)
        m._backward_impl(st, torch.from_numpy(g).to(DEV))
        twin.backward(g)
    _check_rows(m, twin, hi)


@pytest.mark.parametrize("mode", ["hybrid", "cache"])
@pytest.mark.parametrize("pooling", ["SUM", "NONE"])
def test_prefetch_pipeline_through_the_tiers_against_the_dict_twin(mode, pooling):
    """_prefetch_cache_path (batched_dynamicemb_function.py:298-556): prefetch(batch i+1) walks BOTH tiers before
    backward(batch i) -- HBM tier of one bucket per table, so its inserts evict into the host tier all the time -- and the
    rows an outstanding batch resolved stay where they are (pinned) until its backward.  Outputs and final rows equal the
    dict twin's; every pin is released at the end."""
    dims, fmap, F, B, hi, lr = [8, 8], [0, 1, 1], 3, 24, 500, 0.1
    m = _module(dims, fmap, pooling, "SGD", lr, local_hbm=2 * 128 * 4 * dims[0], caching=mode == "cache", prefetch_pipeline=True)
    assert m.storage_mode == "hybrid"
    twin = DictEmbeddingTwin(dims, fmap, pooling, "sgd", lr=lr)
    rng = np.random.default_rng(29)
    batches = [_batch(rng, F, B, hi) for _ in range(8)]
    dev = [(torch.from_numpy(k).to(DEV), torch.from_numpy(o).to(DEV)) for k, o in batches]
    m.prefetch(*dev[0])
    for i, (keys, off) in enumerate(batches):
        out, st = m._forward_impl(*dev[i], train=True)
        if i + 1 < len(batches):
            m.prefetch(*dev[i + 1])
        ref = twin.forward(keys, off, train=True)
        np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=1e-6, atol=1e-3)
        g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)
        m._backward_impl(st, torch.from_numpy(g).to(DEV))
        twin.backward(g)
    _check_rows(m, twin, hi)
    assert int(m.size()) == sum(len(t) for t in twin.tables)
    assert int(m.table_host.size()) > 0, "the HBM tier never spilled"
    assert m._tier_prefetched == 0
    assert int(m.table._ref_counter.sum()) == 0 and int(m.table_host._ref_counter.sum()) == 0
    # a plain (not prefetched) step afterwards still works, and promotion is allowed again
    keys, off = _batch(rng, F, B, hi)
    out, st = m._forward_impl(torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV), train=True)
    np.testing.assert_allclose(out.double().cpu().numpy(), twin.forward(keys, off, train=True), rtol=1e-6, atol=1e-3)


@pytest.fixture(scope="module")
def pg():
    import torch.distributed as dist

    from conftest import rendezvous_file
    dist.init_process_group("nccl", init_method=rendezvous_file(), rank=0, world_size=1, device_id=DEV)
    yield dist.group.WORLD
    dist.destroy_process_group()


def test_pre_communication_dedup_against_the_dict_twin(pg):
    """rows-back pooled mode: local dedup -> exchange of the unique keys -> rows back -> local pooling; backward reduces to
    unique gradients before the exchange"""
    from dynamicemb.sharded import RowWiseShardedPooledRows, _ModuleLocal

    dims, F, B, hi, lr = [16, 16], 2, 48, 300, 0.2
    loc = _module(dims, [0, 1], "NONE", "SGD", lr)
    sh = RowWiseShardedPooledRows(_ModuleLocal(loc), [0, 1], [hi, hi], dims, combiner=0, device=DEV, out_dtype=torch.float32,
                                  dist_type_per_table=["hash_roundrobin"] * 2, chunk=16)
    twin = DictEmbeddingTwin(dims, [0, 1], "SUM", "sgd", lr=lr)
    rng = np.random.default_rng(8)
    for step in range(5):
        keys, off = _batch(rng, F, B, hi, maxlen=9)
        out, ctx = sh.forward(torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV), True)
        ref = twin.forward(keys, off, True)
        np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=1e-6, atol=1e-3)
        g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)
        sh.backward(ctx, torch.from_numpy(g).to(DEV))
        twin.backward(g)
    _check_rows(loc, twin, hi)


@pytest.mark.parametrize("optimizer", ["SGD", "ADAM", "EXACT_ROWWISE_ADAGRAD"])
def test_checkpoint_files_against_the_dict_twin(tmp_path, optimizer):
    """dump() after a few training steps: the raw little-endian files (keys i64 / values f32 / scores i64 / opt_values f32,
    no headers) parsed by numpy hold exactly the twin's rows and optimizer state; a fresh module that load()s them serves
    the twin's rows"""
    dims, fmap, F, B, hi, lr = [8, 16], [0, 1], 2, 24, 500, 0.05
    m = _module(dims, fmap, "SUM", optimizer, lr)
    twin = DictEmbeddingTwin(dims, fmap, "SUM", _OPT[optimizer], lr=lr)
    rng = np.random.default_rng(4)
    for step in range(4):
        keys, off = _batch(rng, F, B, hi)
        out, st = m._forward_impl(torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV), train=True)
        ref = twin.forward(keys, off, True)
        g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)
        m._backward_impl(st, torch.from_numpy(g).to(DEV))
        twin.backward(g)
    m.dump(str(tmp_path), optim=True)
    ckpt_state = {"SGD": lambda d: 0, "ADAM": lambda d: 2 * d, "EXACT_ROWWISE_ADAGRAD": lambda d: 1}[optimizer]
    for t, name in enumerate(m._table_names):
        d = dims[t]
        base = lambda item: os.path.join(tmp_path, f"{name}_emb_{item}.rank_0.world_size_1")  # noqa: E731
        k = np.fromfile(base("keys"), dtype="<i8")
        v = np.fromfile(base("values"), dtype="<f4").reshape(-1, d)
        s = np.fromfile(base("scores"), dtype="<i8")
        assert sorted(k.tolist()) == sorted(twin.tables[t].keys()) and s.size == k.size and v.shape[0] == k.size
        want = np.stack([twin.tables[t][int(x)] for x in k])
        np.testing.assert_allclose(v, want[:, :d], rtol=2e-5, atol=2e-4)
        cs = ckpt_state(d)
        if cs:
            o = np.fromfile(base("opt_values"), dtype="<f4").reshape(-1, cs)
            np.testing.assert_allclose(o, want[:, d:d + cs], rtol=2e-4, atol=1e-5)
        meta = json.load(open(os.path.join(tmp_path, f"{name}_opt_args.json")))
        assert meta["opt_type"] in ("sgd", "adam", "exact_row_wise_adagrad") and abs(meta["lr"] - lr) < 1e-12
    fresh = _module(dims, fmap, "SUM", optimizer, lr)
    fresh.load(str(tmp_path), optim=True)
    _check_rows(fresh, twin, hi)
    # and training continues identically from the loaded state (optimizer state and step count came along)
    keys, off = _batch(rng, F, B, hi)
    out, st = fresh._forward_impl(torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV), train=True)
    ref = twin.forward(keys, off, True)
    np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=1e-6, atol=1e-3)
    g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)
    fresh._backward_impl(st, torch.from_numpy(g).to(DEV))
    twin.backward(g)
    _check_rows(fresh, twin, hi)


def test_table_grows_under_the_prefetch_pipeline_order():
    """The reference's pipeline order -- prefetch(batch k + 1) before forward / backward of batch k (train_pipeline.py:533-692) --
    always has a batch queued when a growth is due (round-4 advisor finding: the table then never grew and evicted silently).
    Now the deferred growth runs at the end of the backward that leaves no forward alive; the queued batch's prefetch is dropped
    and re-resolved against the grown table.  Checked against the dict twin: nothing is evicted, every key keeps its row."""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    dims, fmap, F, B, lr = [8], [0], 1, 96, 0.05
    opts = [DynamicEmbTableOptions(dim=d, init_capacity=256, max_capacity=16384, max_load_factor=0.5, index_type=torch.int64,
                                   embedding_dtype=torch.float32, score_strategy=DynamicEmbScoreStrategy.STEP,
                                   initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG)) for d in dims]
    m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=fmap, pooling_mode=DynamicEmbPoolingMode.SUM,
                                        output_dtype=torch.float32, optimizer=EmbOptimType.SGD, learning_rate=lr, device=DEV)
    m.train()
    twin = DictEmbeddingTwin(dims, fmap, "SUM", "sgd", lr=lr)
    rng = np.random.default_rng(21)
    batches = [_batch(rng, F, B, 150 * (k + 1)) for k in range(26)]
    dev = [(torch.from_numpy(k).to(DEV), torch.from_numpy(o).to(DEV)) for k, o in batches]
    m.prefetch(*dev[0])
    caps_seen = set()
    for k in range(len(dev)):
        if k + 1 < len(dev):
            m.prefetch(*dev[k + 1])
        out, st = m._forward_impl(*dev[k], train=True)
        ref = twin.forward(batches[k][0], batches[k][1], True)
        np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=1e-6, atol=1e-3, err_msg=f"step {k}")
        g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)
        m._backward_impl(st, torch.from_numpy(g).to(DEV))
        twin.backward(g)
        torch.cuda.synchronize()
        caps_seen.add(tuple(m.table.per_table_capacity_))
    assert len(caps_seen) >= 3 and max(caps_seen)[0] >= 2048, caps_seen
    _check_rows(m, twin, 150 * 26)
    assert int(m.size()) == sum(len(t) for t in twin.tables)      # nothing evicted on the way


def test_table_growth_by_rehash_against_the_dict_twin():
    """init_capacity 256 -> max_capacity 16384 (key_value_table.py:559-666): the table doubles by rehash when its fill
    passes max_load_factor; every key keeps its row and optimizer state across the moves, the value buffer keeps its base
    address (VMM), nothing is evicted on the way"""
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    dims, fmap, F, B, lr = [8, 8], [0, 1], 2, 64, 0.05
    opts = [DynamicEmbTableOptions(dim=d, init_capacity=256, max_capacity=16384, max_load_factor=0.5, index_type=torch.int64,
                                   embedding_dtype=torch.float32, score_strategy=DynamicEmbScoreStrategy.STEP,
                                   initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG)) for d in dims]
    m = BatchedDynamicEmbeddingTablesV2(opts, feature_table_map=fmap, pooling_mode=DynamicEmbPoolingMode.SUM,
                                        output_dtype=torch.float32, optimizer=EmbOptimType.ADAM, learning_rate=lr, device=DEV)
    m.train()
    assert m._growth and m.table.per_table_capacity_ == [256, 256]
    base = [v.data_ptr() for v in m.values]
    twin = DictEmbeddingTwin(dims, fmap, "SUM", "adam", lr=lr)
    rng = np.random.default_rng(12)
    caps_seen = set()
    for step in range(24):
        hi = 200 * (step + 1)                       # the key space keeps widening: ~4000 keys per table at the end
        keys, off = _batch(rng, F, B, hi)
        out, st = m._forward_impl(torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV), train=True)
        ref = twin.forward(keys, off, True)
        np.testing.assert_allclose(out.double().cpu().numpy(), ref, rtol=1e-6, atol=1e-3)
        g = rng.uniform(0.1, 1.1, size=ref.shape).astype(np.float32)
        m._backward_impl(st, torch.from_numpy(g).to(DEV))
        twin.backward(g)
        torch.cuda.synchronize()
        caps_seen.add(tuple(m.table.per_table_capacity_))
    assert len(caps_seen) >= 4 and max(caps_seen)[0] >= 4096, caps_seen
    assert [v.data_ptr() for v in m.values] == base
    _check_rows(m, twin, 4800)
    assert int(m.size()) == sum(len(t) for t in twin.tables)
