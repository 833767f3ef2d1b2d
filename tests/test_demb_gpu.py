"""GPU parity tests of the DynamicEmb path: every HIP kernel (through the C ABI / the
dynamicemb_extensions drop-in) against the CPU oracle on identical seeded inputs.  Bit exact for
keys / indices / counts / table bytes; 1e-3 relative (BASELINE.json north_star) for bf16 values,
1e-5 for fp32 values.  Mirrors corelib/dynamicemb/test/test_unique_op.py,
test/unit_tests/table_operation/test_table_operation.py and the kernel-level parts of
test_batched_dynamic_embedding_tables_v2.py of the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ext():
    import dynamicemb_extensions as e

    return e


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def rand_keys(rng, n, hi=1 << 40):
    return rng.choice(hi, size=n, replace=False).astype(np.int64)


def make_tables(caps, C=128, ns=1, policy=None):
    from dynamicemb.scored_hashtable import LinearBucketTable, ScoreSpec

    e = ext()
    pol = policy if policy is not None else e.ScorePolicy.ASSIGN
    g = LinearBucketTable(list(caps), [ScoreSpec("s", pol)], bucket_capacity=C, device=torch.device(DEV))
    o = orc.OracleTable(list(caps), bucket_capacity=C, num_scores=ns)
    return g, o


def storage_np(g):
    return g.table_storage_.cpu().numpy()


# ------------------------------------------------------------------------------------------ table
def test_table_init_bytes():
    g, o = make_tables([1000, 300], C=128)
    assert (storage_np(g) == o.storage).all()
    g2, o2 = make_tables([64], C=16, ns=2, policy=ext().ScorePolicy.LRU_LFU)
    assert (storage_np(g2) == o2.storage).all()


@pytest.mark.parametrize("C", [16, 128, 256, 1024])
def test_deterministic_insert_bit_exact(C, monkeypatch):
    """DEMB_DETERMINISM_MODE waves: slot indices, arena bytes and bucket sizes equal the sequential
    restatement (SURVEY 8(c) parity definition (i))."""
    monkeypatch.setenv("DEMB_DETERMINISM_MODE", "ON")
    from dynamicemb.scored_hashtable import ScoreArg

    e = ext()
    rng = np.random.default_rng(C)
    caps = [8 * C, 3 * C]
    g, o = make_tables(caps, C=C)
    n = int(1.6 * sum(caps))  # overfull: forces evictions
    keys = rand_keys(rng, n)
    tids = (rng.random(n) < 0.3).astype(np.int64)
    sc = rng.integers(1, 1 << 20, size=n).astype(np.int64)
    for lo, hi in ((0, n // 2), (n // 2, n)):
        io = o.insert_deterministic(keys[lo:hi], tids[lo:hi], sc[lo:hi].view(np.uint64), orc.POLICY_ASSIGN)
        ig = g.insert(T(keys[lo:hi]), T(tids[lo:hi]), ScoreArg("s", T(sc[lo:hi]).view(torch.uint64), e.ScorePolicy.ASSIGN))
        assert (ig.cpu().numpy() == io).all()
        assert (storage_np(g) == o.storage).all()
        assert (g.bucket_sizes.cpu().numpy() == o.bucket_sizes).all()
    so_o, f_o, i_o = o.lookup(keys, tids)
    so_g, f_g, i_g = g.lookup(T(keys), T(tids), ScoreArg("s", None, e.ScorePolicy.CONST))
    assert (f_g.cpu().numpy() == f_o).all() and (i_g.cpu().numpy() == i_o).all()
    assert (so_g.cpu().numpy() == so_o).all()
    assert 0 < f_o.sum() < n  # some keys were evicted


_FLOW = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demb_flow_golden.npz"))


@pytest.mark.parametrize("name", [str(x) for x in _FLOW["flow_cases"]])
def test_deterministic_insert_matches_the_reference_python_flow(name, monkeypatch):
    """the fixture holds what the REFERENCE's `_bucketize_and_pad` + `_deterministic_insert` (scored_hashtable.py
    :1451-1558, pulled out of its AST by tests/golden/gen_demb_flow_golden.py) produce over the oracle kernels: the HIP
    table in DEMB_DETERMINISM_MODE must give every key the same slot and leave the same arena bytes"""
    monkeypatch.setenv("DEMB_DETERMINISM_MODE", "ON")
    from dynamicemb.scored_hashtable import LinearBucketTable, ScoreArg, ScoreSpec

    e = ext()
    gg = lambda k: _FLOW[f"{name}/{k}"]
    C, policy, nb = [int(x) for x in gg("C")]
    g = LinearBucketTable([int(c) for c in gg("caps")], [ScoreSpec("s", e.ScorePolicy(policy))], bucket_capacity=C,
                          device=torch.device(DEV))
    for i in range(nb):
        keys = T(gg(f"keys{i}").view(np.int64))
        idx = g.insert(keys, T(gg(f"tids{i}")), ScoreArg("s", T(gg(f"scores{i}").view(np.int64)).view(torch.uint64),
                                                         e.ScorePolicy(policy)))
        assert np.array_equal(idx.cpu().numpy(), gg(f"idx{i}")), f"batch {i}"
    assert np.array_equal(storage_np(g), gg("arena")) and np.array_equal(g.bucket_sizes.cpu().numpy(), gg("bucket_sizes"))


def test_deterministic_insert_and_evict_streams(monkeypatch):
    monkeypatch.setenv("DEMB_DETERMINISM_MODE", "ON")
    from dynamicemb.scored_hashtable import ScoreArg

    e = ext()
    rng = np.random.default_rng(7)
    C = 32
    g, o = make_tables([4 * C], C=C)
    keys = rand_keys(rng, 10 * C)
    tids = np.zeros(keys.size, np.int64)
    sc = rng.integers(1, 1 << 30, size=keys.size).astype(np.int64)
    io, evo = o.insert_deterministic(keys, tids, sc.view(np.uint64), orc.POLICY_ASSIGN, evict_out=True)
    ig, nev, ek, ei, es, et = g.insert_and_evict(T(keys), T(tids), ScoreArg("s", T(sc).view(torch.uint64), e.ScorePolicy.ASSIGN))
    assert (ig.cpu().numpy() == io).all()
    assert nev == evo[0].size
    # order inside a wave is arbitrary on the GPU: compare as multisets
    a = sorted(zip(ek.cpu().numpy().astype(np.uint64).tolist(), ei.cpu().numpy().tolist(), es.cpu().numpy().tolist()))
    b = sorted(zip(evo[0].astype(np.uint64).tolist(), evo[1].tolist(), evo[2].tolist()))
    assert a == b
    assert (storage_np(g) == o.storage).all()


def test_concurrent_insert_invariants():
    """One launch, many keys per bucket: schedule dependent slots, so check the order-free
    invariants of SURVEY 8(c)(ii) + test_table_operation.py: round trip, bucket rule, bijection,
    bucket_sizes, results, nothing evicted while empties exist."""
    from dynamicemb.scored_hashtable import ScoreArg

    e = ext()
    rng = np.random.default_rng(11)
    C = 128
    caps = [64 * C, 16 * C]
    g, _ = make_tables(caps, C=C)
    n = 6000
    keys = rand_keys(rng, n)
    tids = (rng.random(n) < 0.25).astype(np.int64)
    sc = rng.integers(1, 1 << 30, size=n).astype(np.int64)
    res = torch.empty(n, dtype=torch.uint8, device=DEV)
    idx = g.insert(T(keys), T(tids), ScoreArg("s", T(sc).view(torch.uint64), e.ScorePolicy.ASSIGN), insert_results=res)
    idx = idx.cpu().numpy()
    res = res.cpu().numpy()
    assert (res == orc.RES_INSERT).all() and (idx >= 0).all()
    so, f, i2 = g.lookup(T(keys), T(tids), ScoreArg("s", None, e.ScorePolicy.CONST))
    assert f.all().item() and (i2.cpu().numpy() == idx).all() and (so.cpu().numpy() == sc).all()
    for t in (0, 1):
        m = tids == t
        assert np.unique(idx[m]).size == m.sum()
        cap = caps[t]
        exp_bucket = np.array([(orc.hash64(int(k)) % cap) // C for k in keys[m][:300]])
        assert (idx[m][:300] // C == exp_bucket).all()
    assert int(g.size().item()) == n
    # re-insert the same keys: ASSIGN, same slots, new scores
    res2 = torch.empty(n, dtype=torch.uint8, device=DEV)
    idx2 = g.insert(T(keys), T(tids), ScoreArg("s", T(sc + 5).view(torch.uint64), e.ScorePolicy.ASSIGN), insert_results=res2)
    assert (res2.cpu().numpy() == orc.RES_ASSIGN).all() and (idx2.cpu().numpy() == idx).all()
    so, _, _ = g.lookup(T(keys), T(tids), ScoreArg("s", None, e.ScorePolicy.CONST))
    assert (so.cpu().numpy() == sc + 5).all()
    assert int(g.size().item()) == n


def test_concurrent_overfull_eviction_invariants():
    from dynamicemb.scored_hashtable import ScoreArg

    e = ext()
    rng = np.random.default_rng(13)
    C = 128
    g, _ = make_tables([4 * C], C=C)
    old = rand_keys(rng, 4 * C * 4)
    z = lambda n: torch.zeros(n, dtype=torch.int64, device=DEV)
    # fill completely with low scores (several launches so everything lands)
    for _ in range(3):
        g.insert(T(old), z(old.size), ScoreArg("s", torch.full((old.size,), 5, dtype=torch.int64, device=DEV).view(torch.uint64), e.ScorePolicy.ASSIGN))
    assert int(g.size().item()) == 4 * C
    new = rand_keys(rng, 200, hi=1 << 41) + (1 << 41)
    res = torch.empty(200, dtype=torch.uint8, device=DEV)
    idx, nev, ek, ei, es, et = g.insert_and_evict(T(new), z(200), ScoreArg("s", torch.full((200,), 99, dtype=torch.int64, device=DEV).view(torch.uint64), e.ScorePolicy.ASSIGN), insert_results=res)
    res = res.cpu().numpy()
    assert ((res == orc.RES_EVICT) | (res == orc.RES_BUSY)).all()
    n_ev = int((res == orc.RES_EVICT).sum())
    assert n_ev > 0 and nev == 200
    assert (es.cpu().numpy()[ei.cpu().numpy() >= 0] == 5).all()          # only old (score 5) entries evicted
    _, f, i2 = g.lookup(T(new), z(200), ScoreArg("s", None, e.ScorePolicy.CONST))
    assert (f.cpu().numpy() == (res == orc.RES_EVICT)).all()
    assert int(g.size().item()) == 4 * C
    # evicted old keys are gone
    evk = ek.cpu().numpy()[ei.cpu().numpy() >= 0]
    _, f3, _ = g.lookup(T(evk), z(evk.size), ScoreArg("s", None, e.ScorePolicy.CONST))
    assert not f3.any().item()


def test_policies_erase_counter_reserved(monkeypatch):
    from dynamicemb.scored_hashtable import ScoreArg

    e = ext()
    C = 16
    g, o = make_tables([C], C=C)
    z = lambda n: torch.zeros(n, dtype=torch.int64, device=DEV)
    k = T(np.arange(1, C + 1, dtype=np.int64))
    monkeypatch.setenv("DEMB_DETERMINISM_MODE", "ON")
    sc = np.arange(10, 10 + C).astype(np.int64)
    ig = g.insert(k, z(C), ScoreArg("s", T(sc).view(torch.uint64), e.ScorePolicy.ASSIGN))
    io = o.insert_deterministic(np.arange(1, C + 1, dtype=np.int64), np.zeros(C, np.int64), sc.view(np.uint64), orc.POLICY_ASSIGN)
    assert (ig.cpu().numpy() == io).all()
    monkeypatch.delenv("DEMB_DETERMINISM_MODE")
    # ACCUMULATE on lookup
    so, f, _ = g.lookup(k[:4], z(4), ScoreArg("s", T(np.full(4, 3, np.int64)).view(torch.uint64), e.ScorePolicy.ACCUMULATE))
    so_o, _, _ = o.lookup(np.arange(1, 5, dtype=np.int64), np.zeros(4, np.int64), np.full(4, 3, np.uint64), orc.POLICY_ACCUMULATE)
    assert (so.cpu().numpy() == so_o).all() and f.all().item()
    # GLOBAL_TIMER with the test hook
    e.TIMER_OVERRIDE = 777777
    so, _, _ = g.lookup(k[:2], z(2), ScoreArg("s", None, e.ScorePolicy.GLOBAL_TIMER))
    e.TIMER_OVERRIDE = 0
    o.lookup(np.arange(1, 3, dtype=np.int64), np.zeros(2, np.int64), None, orc.POLICY_GLOBAL_TIMER, timer=777777)
    assert (so.cpu().numpy() == 777777).all()
    assert (storage_np(g) == o.storage).all()
    # real device clock is monotonic
    t0 = e.device_timestamp()
    t1 = e.device_timestamp()
    assert t1 >= t0 > 0
    # erase -> tombstones, then reclaim lowest slot first
    g.erase(k[:3], z(3))
    o.erase(np.arange(1, 4, dtype=np.int64), np.zeros(3, np.int64))
    assert (storage_np(g) == o.storage).all() and (g.bucket_sizes.cpu().numpy() == o.bucket_sizes).all()
    res = torch.empty(1, dtype=torch.uint8, device=DEV)
    i2 = g.insert(T(np.array([77], np.int64)), z(1), ScoreArg("s", T(np.array([9], np.int64)).view(torch.uint64), e.ScorePolicy.ASSIGN), insert_results=res)
    io2, ro2, _ = o.insert(np.array([77], np.int64), np.zeros(1, np.int64), np.array([9], np.uint64), orc.POLICY_ASSIGN)
    assert res.item() == orc.RES_RECLAIM == ro2[0] and i2.item() == io2[0]
    assert (storage_np(g) == o.storage).all()
    # pinned slots are never evicted
    g.increment_counter(torch.arange(C, device=DEV), z(C))
    o.counter[:] = 1
    res = torch.empty(1, dtype=torch.uint8, device=DEV)
    i3 = g.insert(T(np.array([31337], np.int64)), z(1), ScoreArg("s", T(np.array([1], np.int64)).view(torch.uint64), e.ScorePolicy.ASSIGN), insert_results=res)
    assert res.item() == orc.RES_BUSY and i3.item() == -1
    g.decrement_counter(torch.arange(C, device=DEV), z(C))
    assert int(g._ref_counter.abs().sum().item()) == 0
    # reserved keys
    bad = torch.tensor([-1, -2, -3, -4], dtype=torch.int64, device=DEV)
    res = torch.empty(4, dtype=torch.uint8, device=DEV)
    i4 = g.insert(bad, z(4), ScoreArg("s", T(np.ones(4, np.int64)).view(torch.uint64), e.ScorePolicy.ASSIGN), insert_results=res)
    assert (i4 == -1).all().item() and (res == orc.RES_ILLEGAL).all().item()
    _, f4, i5 = g.lookup(bad, z(4), ScoreArg("s", None, e.ScorePolicy.CONST))
    assert not f4.any().item() and (i5 == -1).all().item()


def test_lru_lfu_two_word_scores(monkeypatch):
    from dynamicemb.scored_hashtable import ScoreArg

    e = ext()
    monkeypatch.setenv("DEMB_DETERMINISM_MODE", "ON")
    g, o = make_tables([64], C=16, ns=2, policy=e.ScorePolicy.LRU_LFU)
    monkeypatch.delenv("DEMB_DETERMINISM_MODE")
    rng = np.random.default_rng(5)
    keys = rand_keys(rng, 40)
    z = np.zeros(40, np.int64)
    fr = rng.integers(1, 10, size=40).astype(np.int64)
    e.TIMER_OVERRIDE = 1000
    # unique buckets not guaranteed -> use the oracle in "one call" mode only for non-colliding subset: insert one by one
    for i in range(40):
        ig = g.insert(T(keys[i:i + 1]), T(z[:1]), ScoreArg("s", T(fr[i:i + 1]).view(torch.uint64), e.ScorePolicy.LRU_LFU))
        io, _, _ = o.insert(keys[i:i + 1], z[:1], fr[i:i + 1].view(np.uint64), orc.POLICY_LRU_LFU, timer=1000)
        assert ig.item() == io[0]
    e.TIMER_OVERRIDE = 2000
    so, f, _ = g.lookup(T(keys), T(z), ScoreArg("s", T(fr).view(torch.uint64), e.ScorePolicy.LRU_LFU))
    so_o, f_o, _ = o.lookup(keys, z, fr.view(np.uint64), orc.POLICY_LRU_LFU, timer=2000)
    e.TIMER_OVERRIDE = 0
    assert (so.cpu().numpy() == so_o).all() and (f.cpu().numpy() == f_o).all()
    assert (storage_np(g) == o.storage).all()


# --------------------------------------------------------------------------------------- index ops
@pytest.mark.parametrize("n,T_,zipf", [(0, 2, False), (1, 1, False), (5000, 3, False), (200000, 4, True), (65536, 1, True)])
def test_segmented_unique_exact(n, T_, zipf):
    e = ext()
    rng = np.random.default_rng(n + T_)
    cuts = np.sort(rng.integers(0, n + 1, size=T_ - 1)) if T_ > 1 else np.array([], np.int64)
    seg = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    if T_ >= 3 and n > 10:
        seg[2] = seg[1]  # an empty table in the middle
    if zipf:
        keys = (rng.zipf(1.3, size=n) % 50000).astype(np.int64) * 7919
    else:
        keys = rng.integers(0, max(n // 3, 2), size=n).astype(np.int64)
    freq_in = rng.integers(1, 5, size=n).astype(np.int64)
    uk, oi, to, fr = orc.segmented_unique(keys, seg, in_freq=freq_in, count_freq=True)
    num, guk, goi, gto, gfr = e.segmented_unique_cuda(T(keys), T(seg), T_, T(freq_in))
    nu = int(num.item())
    assert nu == uk.size
    assert (gto.cpu().numpy() == to).all()
    assert (guk.cpu().numpy()[:nu].view(np.uint64) == uk).all()
    assert (goi.cpu().numpy() == oi).all()
    assert (gfr.cpu().numpy()[:nu] == fr).all()
    # reference invariants (test_unique_op.py:81-105)
    if n:
        assert (guk[goi] == T(keys)).all().item()
    # counting without input frequencies and without counting
    _, _, goi2, _, gfr2 = e.segmented_unique_cuda(T(keys), T(seg), T_, torch.empty(0, dtype=torch.int64, device=DEV))
    _, _, _, fr2 = orc.segmented_unique(keys, seg, count_freq=True)
    assert (gfr2.cpu().numpy()[:nu] == fr2).all() and (goi2.cpu().numpy() == oi).all()
    _, _, goi3, _, gfr3 = e.segmented_unique_cuda(T(keys), T(seg), T_, None)
    assert gfr3.numel() == 0 and (goi3.cpu().numpy() == oi).all()


def test_expand_table_ids_table_range_compact():
    e = ext()
    off = np.array([0, 5, 5, 9, 20], np.int64)
    got = e.expand_table_ids_cuda(T(off), 20).cpu().numpy()
    assert (got == orc.expand_table_ids(off, 20)).all()
    offsets = np.array([0, 2, 3, 5, 6, 8, 10, 10, 11], np.int64)
    fo = np.array([0, 2, 3, 4], np.int64)
    assert e.get_table_range(T(offsets), T(fo)).cpu().tolist() == [0, 6, 10, 11]
    rng = np.random.default_rng(2)
    for n in (0, 1, 1023, 1024, 5000, 70000):
        flags = rng.random(n) < 0.3
        a = rng.integers(0, 1 << 40, size=n).astype(np.int64)
        b = rng.integers(0, 8, size=n).astype(np.int64)
        cnt, idx, (ca, cn, cb) = e.flagged_compact(T(flags), [T(a), None, T(b)])
        assert cnt == flags.sum() and cn is None
        assert (idx.cpu().numpy() == np.nonzero(flags)[0]).all()
        assert (ca.cpu().numpy() == a[flags]).all() and (cb.cpu().numpy() == b[flags]).all()


@pytest.mark.parametrize("W", [1, 3, 8])
def test_block_bucketize_positions(W):
    """bucketize_pos=True: new_pos[j] = the position the j-th bucketized value had inside its original bag
    (reference: block_bucketize_sparse_features, sparse_block_bucketize_features.cu:366-830, called with bucketize_pos by
    input_dist.py:140-155 for position-weighted features); the permutation is returned only when asked for."""
    e = ext()
    rng = np.random.default_rng(W + 40)
    F, B = 2, 7
    lens = rng.integers(0, 40, size=F * B)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, 1000, size=int(offsets[-1])).astype(np.int64)
    blk = np.array([1000 // W + 1] * F, np.int64)
    nl, ni, _, pos, perm = e.block_bucketize_sparse_features(T(lens.astype(np.int64)), T(idx), True, True, T(np.array([0, 1], np.int32)), T(blk), W)
    nl2, ni2, _, pos2, perm2 = e.block_bucketize_sparse_features(T(lens.astype(np.int64)), T(idx), True, False, T(np.array([0, 1], np.int32)), T(blk), W)
    assert perm2 is None and torch.equal(pos, pos2) and torch.equal(ni, ni2) and torch.equal(nl, nl2)
    perm, pos, ni = perm.cpu().numpy(), pos.cpu().numpy(), ni.cpu().numpy()
    bag = np.repeat(np.arange(F * B), lens)
    within = np.arange(idx.size) - offsets[bag]
    assert sorted(perm.tolist()) == list(range(idx.size))   # value j went to place perm[j] (its new value is the shard-local id) ...
    assert (pos[perm] == within).all()                      # ... and took its in-bag position along


@pytest.mark.parametrize("W", [1, 3, 8, 70])
@pytest.mark.parametrize("many_bags", [False, True], ids=["wave_per_bag", "lane_group_per_bag"])
def test_block_bucketize_variable_batch_and_uneven_shard_boundaries(W, many_bags):
    """the two remaining arguments of the reference's op (sparse_block_bucketize_features.cu:194-211, 262-292, 341-347):
    batch_size_per_feature (features with different batch sizes: a bag's feature comes from the prefix sum of the sizes) and
    block_bucketize_pos (uneven shard boundaries: rank = last boundary <= idx, new index = idx - boundary; indices outside the
    boundaries fall back to idx % W, idx / W) -- each alone and both together, both kernel families, against the oracle;
    bucketize_pos rides along."""
    e = ext()
    rng = np.random.default_rng(W * 2 + int(many_bags))
    bs = np.array([5, 0, 9, 2], np.int64) * (1200 if many_bags else 1)     # feature 1 has no bag at all
    F = bs.size
    FB = int(bs.sum())
    lens = rng.integers(0, 6 if many_bags else 150, size=FB)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, 5000, size=int(offsets[-1])).astype(np.int64)
    blk = np.array([5000 // W + 1] * F, np.int64)
    bag_feature = np.repeat(np.arange(F), bs)
    # boundaries: W + 1 sorted cut points per feature; the first one above 0 and the last below 5000 for features 0 / 2, so that
    # indices on either side take the fallback
    pos = []
    for f in range(F):
        cuts = np.sort(rng.choice(np.arange(1, 4999), size=W - 1, replace=False)) if W > 1 else np.zeros(0, np.int64)
        lo, hi = (40, 4900) if f % 2 == 0 else (0, 5000)
        pos.append(np.concatenate([[lo], np.clip(cuts, lo, hi), [hi]]).astype(np.int64))
        pos[-1].sort()
    dist = np.array([0, 1, 2, 0], np.int32)
    for use_bs, use_pos in ((True, False), (False, True), (True, True)):
        if not use_bs:      # equal batch sizes: re-draw the bags as F x B
            B = FB // F
            lens_c = lens[: F * B]
            off_c = np.concatenate([[0], np.cumsum(lens_c)]).astype(np.int64)
            idx_c = idx[: off_c[-1]]
            bf = None
        else:
            B, lens_c, off_c, idx_c, bf = 0, lens, offsets, idx, bag_feature
        nFB = lens_c.size
        onl, ono, oni, operm = orc.block_bucketize_ex(off_c, idx_c, W, max(B, 1), blk, dist, bf, pos if use_pos else None)
        gl, gi, _, gpos, gperm = e.block_bucketize_sparse_features(
            T(lens_c.astype(np.int64)), T(idx_c), True, True, T(dist), T(blk), W,
            batch_size_per_feature=T(bs) if use_bs else None, block_bucketize_pos=[T(x) for x in pos] if use_pos else None)
        assert (gl.cpu().numpy() == onl).all()
        assert (gi.cpu().numpy() == oni.view(np.int64)).all()
        assert (gperm.cpu().numpy() == operm).all()
        bag = np.repeat(np.arange(nFB), lens_c)
        assert (gpos.cpu().numpy()[operm] == np.arange(idx_c.size) - off_c[bag]).all()


@pytest.mark.parametrize("W", [1, 2, 8, 70])
def test_block_bucketize_exact(W):
    e = ext()
    rng = np.random.default_rng(W)
    F, B = 3, 5
    lens = rng.integers(0, 300, size=F * B)
    lens[3] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, 4000, size=int(offsets[-1])).astype(np.int64)
    blk = np.array([4000 // W + 1, 37, 4000 // W + 1], np.int64)
    for dist in ([1, 1, 1], [2, 2, 2], [0, 0, 0], [0, 1, 2]):
        nl, no, ni, perm = None, None, None, None
        # the oracle takes one dist type per call: run per feature and stitch by comparing per-feature
        gl, gi, _, _, gperm = e.block_bucketize_sparse_features(T(lens.astype(np.int64)), T(idx), False, True,
                                                                 T(np.array(dist, np.int32)), T(blk), W)
        gl = gl.cpu().numpy(); gi = gi.cpu().numpy(); gperm = gperm.cpu().numpy()
        assert gl.sum() == idx.size and sorted(gperm.tolist()) == list(range(idx.size))
        go = np.concatenate([[0], np.cumsum(gl)])
        for f in range(F):
            onl, ono, oni, operm = orc.block_bucketize(offsets, idx, W, B, blk, dist[f])
            for b in range(B):
                bag = f * B + b
                for p in range(W):
                    seg_o = oni[ono[p * F * B + bag]: ono[p * F * B + bag + 1]]
                    seg_g = gi[go[p * F * B + bag]: go[p * F * B + bag + 1]]
                    assert (seg_o.view(np.int64) == seg_g).all()
        # unbucketize_permute inverts the routing
        rank_of = np.searchsorted(go, gperm, side="right") - 1
        assert ((rank_of % (F * B)) == np.repeat(np.arange(F * B), lens)).all()


# --------------------------------------------------------------------------------------- value ops
def _tol(dtype):
    return dict(rtol=1e-3, atol=1e-3) if dtype != torch.float32 else dict(rtol=1e-5, atol=1e-5)


def assert_close_lowp(got, exp, dtype, rtol=1e-3, atol=1e-6):
    """Tolerance for 16-bit outputs (north_star: 1e-3 relative on bf16 values): the fp32 sums of both
    sides agree to `rtol`; after the final rounding a value that sits on a rounding boundary may land on
    the neighbouring 16-bit number, so one unit in the last place of the output type is allowed on top."""
    got = np.asarray(got, np.float32)
    exp = np.asarray(exp, np.float32)
    if dtype == torch.float32:
        np.testing.assert_allclose(got, exp, rtol=1e-5, atol=max(1e-5, atol))
        return
    mant = 8 if dtype == torch.bfloat16 else 11
    ulp = np.exp2(np.floor(np.log2(np.maximum(np.abs(exp), 1e-30))) - (mant - 1)).astype(np.float32)
    bad = np.abs(got - exp) > rtol * np.abs(exp) + ulp + atol
    assert not bad.any(), f"{bad.sum()} of {bad.size} beyond 1e-3 rel + 1 ulp; worst {np.abs(got - exp).max()}"


_NP = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}


@pytest.mark.parametrize("D", [8, 7, 32, 128, 256, 512, 13])
@pytest.mark.parametrize("sdt,ddt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16), (torch.float16, torch.float32)])
@pytest.mark.parametrize("combiner", [0, 1])
def test_gather_pooled(D, sdt, ddt, combiner):
    e = ext()
    rng = np.random.default_rng(D * 7 + combiner)
    F, B, Nu = 3, 37, 211
    lens = rng.integers(0, 9, size=F * B)
    lens[5] = 0
    lens[7] = 70  # longer than one unrolled sweep
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rev = rng.integers(0, Nu, size=int(offsets[-1])).astype(np.int64)
    src = orc.round_to(rng.standard_normal((Nu, D)).astype(np.float32), _NP[sdt])
    exp = orc.gather_pooled(src, rev, offsets, B, combiner, out_dtype=_NP[ddt])
    out = torch.empty(B, F * D, dtype=ddt, device=DEV)
    e.gather_embedding_pooled(T(src, sdt), out, T(rev), T(offsets), combiner, F * D, B)
    assert_close_lowp(out.float().cpu().numpy(), exp, ddt)


def test_gather_pooled_mixed_dims_and_row_addr():
    e = ext()
    rng = np.random.default_rng(3)
    dims = [8, 16, 32]
    F, B, Nu = 3, 11, 50
    maxD = 32
    Doff = np.concatenate([[0], np.cumsum(dims)]).astype(np.int32)
    lens = rng.integers(0, 6, size=F * B)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rev = rng.integers(0, Nu, size=int(offsets[-1])).astype(np.int64)
    src = rng.standard_normal((Nu, maxD)).astype(np.float32)
    exp = orc.gather_pooled(src, rev, offsets, B, 0, D_offsets=Doff)
    out = torch.empty(B, int(Doff[-1]), dtype=torch.float32, device=DEV)
    e.gather_embedding_pooled(T(src), out, T(rev), T(offsets), 0, int(Doff[-1]), B, T(Doff), maxD)
    np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)
    # fused form: pool straight from "table" rows through row addresses (value_dim > emb_dim, slot -1 = zeros)
    D, vdim, cap = 16, 48, 300
    table = torch.randn(cap, vdim, device=DEV)
    slots = rng.choice(cap, size=Nu, replace=False).astype(np.int64)
    slots[4] = -1
    tptr = torch.tensor([table.data_ptr()], dtype=torch.int64, device=DEV)
    vd = torch.tensor([vdim], dtype=torch.int64, device=DEV)
    addr = e.row_addresses(T(slots), None, tptr, vd, 4)
    F2 = 2
    lens2 = rng.integers(0, 7, size=F2 * B)
    off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int64)
    rev2 = rng.integers(0, Nu, size=int(off2[-1])).astype(np.int64)
    uniq = orc.gather_rows(table.cpu().numpy(), slots, D)
    exp2 = orc.gather_pooled(uniq, rev2, off2, B, 1, out_dtype="bf16")
    out2 = torch.empty(B, F2 * D, dtype=torch.bfloat16, device=DEV)
    e.gather_embedding_pooled(None, out2, T(rev2), T(off2), 1, F2 * D, B, max_D=D, row_addr=addr, src_dtype=torch.float32)
    assert_close_lowp(out2.float().cpu().numpy(), exp2, torch.bfloat16)


@pytest.mark.parametrize("D", [8, 7, 128, 300])
def test_gather_sequence_and_flat_copy(D):
    e = ext()
    rng = np.random.default_rng(D)
    Nu, N = 97, 1000
    src = rng.standard_normal((Nu, D)).astype(np.float32)
    rev = rng.integers(0, Nu, size=N).astype(np.int64)
    out = torch.empty(N, D, dtype=torch.bfloat16, device=DEV)
    e.gather_embedding(T(src), out, T(rev))
    np.testing.assert_array_equal(out.float().cpu().numpy(), orc.gather_sequence(src, rev, "bf16"))
    # flat tables: two tables with different value dims (emb | optimizer state)
    edims, vdims, caps = [D, max(D // 2, 1)], [D * 3, max(D // 2, 1) * 2], [64, 32]
    tabs = [torch.randn(c, v, device=DEV) for c, v in zip(caps, vdims)]
    tptr = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=DEV)
    tv = torch.tensor(vdims, dtype=torch.int64, device=DEV)
    te = torch.tensor(edims, dtype=torch.int64, device=DEV)
    n = 50
    tid = rng.integers(0, 2, size=n).astype(np.int64)
    idx = np.array([rng.integers(0, caps[t]) for t in tid]).astype(np.int64)
    idx[3] = -1
    maxE, maxV = max(edims), max(edims) + max(v - d for v, d in zip(vdims, edims))
    o_emb = torch.full((n, maxE), -7.0, device=DEV)
    e.load_from_flat_table_emb(tptr, T(idx), T(tid), o_emb, tv, te, maxE, False)
    o_val = torch.full((n, maxV), -7.0, device=DEV)
    e.load_from_flat_table_value(tptr, T(idx), T(tid), o_val, tv, te, maxE, False)
    for i in range(n):
        t = tid[i]
        if idx[i] < 0:
            assert (o_emb[i] == -7).all().item() and (o_val[i] == -7).all().item()
            continue
        row = tabs[t][idx[i]]
        assert torch.equal(o_emb[i, :edims[t]], row[:edims[t]])
        assert torch.equal(o_val[i, :edims[t]], row[:edims[t]])
        assert torch.equal(o_val[i, maxE:maxE + vdims[t] - edims[t]], row[edims[t]:])
    # store back modified values and reload
    uniq_rows = {}
    new_val = torch.randn(n, maxV, device=DEV)
    keep = np.ones(n, bool)
    seen = set()
    for i in range(n):  # unique (table,row) pairs only: stores of duplicates race by design
        if (tid[i], idx[i]) in seen or idx[i] < 0:
            keep[i] = False
        seen.add((tid[i], idx[i]))
    sel = np.nonzero(keep)[0]
    e.store_to_flat_table_value(tptr, T(idx[sel]), T(tid[sel]), new_val[sel].contiguous(), tv, te, maxE, False)
    back = torch.zeros(sel.size, maxV, device=DEV)
    e.load_from_flat_table_value(tptr, T(idx[sel]), T(tid[sel]), back, tv, te, maxE, False)
    for j, i in enumerate(sel):
        t = tid[i]
        assert torch.equal(back[j, :edims[t]], new_val[i, :edims[t]])
        assert torch.equal(back[j, maxE:maxE + vdims[t] - edims[t]], new_val[i, maxE:maxE + vdims[t] - edims[t]])


def test_initializers():
    e = ext()
    keys = torch.tensor([5, 100007, 2**40 + 3], dtype=torch.int64, device=DEV)
    buf = torch.zeros(3, 16, device=DEV)
    e.debug_init(buf, torch.arange(3, device=DEV), keys)
    assert (buf.cpu().numpy() == orc.debug_init(keys.cpu().numpy(), 16)).all()
    e.const_init(buf, torch.tensor([1], device=DEV), 2.5)
    assert (buf[1] == 2.5).all().item() and buf[0, 0].item() == 5.0
    big = torch.zeros(4096, 64, device=DEV)
    ctx = e.CurandStateContext(42)
    kk = torch.arange(4096, device=DEV)
    e.uniform_init(big, None, ctx, -0.5, 0.5, keys=kk)
    assert -0.5 <= big.min().item() and big.max().item() <= 0.5 and abs(big.mean().item()) < 0.01
    assert abs(big.std().item() - (1 / 12) ** 0.5) < 0.01
    big2 = torch.zeros_like(big)
    e.uniform_init(big2, None, ctx, -0.5, 0.5, keys=kk)
    assert torch.equal(big, big2)  # counter based: reproducible per (seed, key)
    e.normal_init(big, None, ctx, 1.0, 2.0, keys=kk)
    assert abs(big.mean().item() - 1.0) < 0.03 and abs(big.std().item() - 2.0) < 0.03
    e.truncated_normal_init(big, None, ctx, 0.0, 1.0, -1.0, 1.0, keys=kk)
    assert big.min().item() >= -1.0 and big.max().item() <= 1.0


# ---------------------------------------------------------------------------------------- backward
def _pooled_case(rng, F, B, Nu, D, hot=0):
    lens = rng.integers(0, 7, size=F * B)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = int(offsets[-1])
    rev = rng.integers(0, Nu, size=n).astype(np.int64)
    if hot:
        rev[rng.choice(n, size=min(hot, n), replace=False)] = 3
    return offsets, rev


@pytest.mark.parametrize("D,gdt", [(8, torch.float32), (7, torch.float32), (128, torch.bfloat16), (256, torch.float32), (24, torch.float16)])
@pytest.mark.parametrize("combiner", [0, 1])
def test_reduce_grads_pooled(D, gdt, combiner):
    e = ext()
    rng = np.random.default_rng(D + combiner)
    F, B, Nu = 2, 300, 150
    offsets, rev = _pooled_case(rng, F, B, Nu, D, hot=700)  # row 3 takes the chunked hot path (> 256 occurrences)
    g = orc.round_to(rng.standard_normal((B, F * D)).astype(np.float32), _NP[gdt])
    exp = orc.reduce_grads_pooled_fast(rev, g, Nu, B, offsets, combiner, out_dtype=_NP[gdt])
    got = e.reduce_grads(T(rev), T(g, gdt), Nu, B, D, T(offsets), None, combiner, F * D)
    # 1e-3 relative + one unit in the last place of the gradient dtype (north star); the absolute term is the fp32
    # accumulation bound of a cancelling sum: 2^-23 x (<= 700 terms) x (|g| <= ~4.5)
    assert_close_lowp(got.float().cpu().numpy(), exp, gdt, atol=4e-4 if gdt == torch.float32 else 4e-4)


def test_reduce_grads_sequence_and_mixed():
    e = ext()
    rng = np.random.default_rng(9)
    n, Nu, D = 5000, 300, 64
    rev = rng.integers(0, Nu, size=n).astype(np.int64)
    rev[:900] = 7
    g = rng.standard_normal((n, D)).astype(np.float32)
    exp = orc.reduce_grads(rev, g, Nu)
    got = e.reduce_grads(T(rev), T(g), Nu, n, D)
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=1e-4, atol=1e-4)
    dims = [8, 16, 32]
    Doff = np.concatenate([[0], np.cumsum(dims)]).astype(np.int32)
    F, B = 3, 20
    offsets, rev2 = _pooled_case(rng, F, B, 40, 32)
    g2 = rng.standard_normal((B, int(Doff[-1]))).astype(np.float32)
    exp2 = orc.reduce_grads(rev2, g2, 40, B, offsets, Doff, 1)
    got2 = e.reduce_grads(T(rev2), T(g2), 40, B, 32, T(offsets), T(Doff), 1, int(Doff[-1]))
    # columns beyond a feature's own width are never written by either side for rows of narrower tables
    np.testing.assert_allclose(got2.cpu().numpy()[:, :8], exp2[:, :8], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("opt", ["sgd", "adam", "adagrad", "rowwise_adagrad"])
@pytest.mark.parametrize("D,wdt,gdt", [(128, torch.float32, torch.bfloat16), (8, torch.float32, torch.float32),
                                       (7, torch.float32, torch.float32), (64, torch.bfloat16, torch.bfloat16)])
def test_backward_fused_optimizers(opt, D, wdt, gdt):
    """fused reduce + in-place optimizer == oracle reduce_grads (rounded to the grad dtype) followed by
    the restated optimizer maths (optimizer_kernel.cuh), 3 iterations."""
    e = ext()
    rng = np.random.default_rng(D)
    F, B, Nu, cap = 2, 400, 120, 500
    kinds = {"sgd": (1, 0), "adam": (2, 2 * D), "adagrad": (3, D), "rowwise_adagrad": (4, 16 // (4 if wdt == torch.float32 else 2))}
    kind, nstate = kinds[opt]
    vdim = D + nstate
    table0 = rng.standard_normal((cap, vdim)).astype(np.float32) * 0.1
    table0[:, D:] = 0.0 if opt != "adagrad" else 0.1
    table0 = orc.round_to(table0, _NP[wdt])
    table = T(table0, wdt).contiguous()
    slots = rng.choice(cap, size=Nu, replace=False).astype(np.int64)
    slots[5] = -1  # failed insert: skipped
    tptr = torch.tensor([table.data_ptr()], dtype=torch.int64, device=DEV)
    addr = e.row_addresses(T(slots), None, tptr, torch.tensor([vdim], dtype=torch.int64, device=DEV), 4 if wdt == torch.float32 else 2)
    hp = dict(lr=0.05, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01)

    def oracle_step(rows, ug, it):
        rows = rows.copy()
        if opt == "sgd":
            orc.sgd_update(rows, ug, D, hp["lr"])
        elif opt == "adam":
            orc.adam_update(rows, ug, D, hp["lr"], hp["beta1"], hp["beta2"], hp["eps"], hp["weight_decay"], it)
        elif opt == "adagrad":
            orc.adagrad_update(rows, ug, D, hp["lr"], hp["eps"])
        else:
            orc.rowwise_adagrad_update(rows, ug, D, hp["lr"], hp["eps"])
        return orc.round_to(rows, _NP[wdt])

    mant_g = {torch.float32: 24, torch.bfloat16: 8, torch.float16: 11}[gdt]
    ok = slots >= 0
    for it in range(1, 4):
        offsets, rev = _pooled_case(rng, F, B, Nu, D, hot=600)
        g = orc.round_to(rng.standard_normal((B, F * D)).astype(np.float32), _NP[gdt])
        ug = orc.reduce_grads_pooled_fast(rev, g, Nu, B, offsets, 1, out_dtype=_NP[gdt])[ok]
        before = table.float().cpu().numpy()       # every iteration starts from the device's own state: no drift
        ptr_t, csr, hot = e.group_by_unique(T(rev), Nu, T(offsets), dim=D)
        e.backward_fused(ptr_t, csr, rev.size, Nu, T(g, gdt), B, D, 1, T(offsets), None, addr, wdt, kind, iter_num=it,
                         state_offset=D, hot=hot, **hp)
        got = table.float().cpu().numpy()
        # The kernel rounds the reduced gradient to the gradient dtype once (as the reference's reduce_grads returns it).
        # Its fp32 sum runs in another order than the oracle's, so an element on a rounding boundary may land on the
        # neighbouring value: the device result must lie between the oracle updates for the reduced gradient one unit in
        # the last place (of the GRADIENT dtype) below and above -- and match within 1e-3 rel + 1 ulp of the WEIGHT dtype
        # (1e-5 for fp32 weights) there.
        step = np.exp2(np.floor(np.log2(np.maximum(np.abs(ug), 1e-30))) - (mant_g - 1)).astype(np.float32)
        if opt != "rowwise_adagrad":
            variants = [oracle_step(before[slots[ok]], ug + d * step, it) for d in (-1.0, 0.0, 1.0)]
        else:
            # row-wise AdaGrad couples the elements of a row through G += mean(g^2): the extremes of G (every element one
            # unit smaller / larger in magnitude) combine with the extremes of the element's own gradient
            b0 = before[slots[ok]]
            variants = [oracle_step(b0, ug, it)]
            for gsign in (-1.0, 1.0):
                G = b0[:, D] + ((np.abs(ug) + gsign * step) ** 2).sum(1, dtype=np.float32) / np.float32(D)
                for d in (-1.0, 1.0):
                    v = b0.copy()
                    v[:, D] = G
                    v[:, :D] = b0[:, :D] - np.float32(hp["lr"]) * (ug + d * step) / (np.sqrt(G)[:, None] + np.float32(hp["eps"]))
                    variants.append(orc.round_to(v, _NP[wdt]))
        lo, hi = np.minimum.reduce(variants), np.maximum.reduce(variants)
        x = got[slots[ok]]
        nearest = np.clip(x, lo, hi)
        assert_close_lowp(x, nearest, wdt, atol=2e-6)
        rest = np.setdiff1d(np.arange(cap), slots[ok])
        assert (got[rest] == before[rest]).all()
    got = table.float().cpu().numpy()
    untouched = np.setdiff1d(np.arange(cap), slots[slots >= 0])
    assert (got[untouched] == table0[untouched]).all()


def test_optimizer_update_ops_dense_grads():
    e = ext()
    rng = np.random.default_rng(21)
    D, N, cap = 32, 40, 64
    for name, kind, nstate in (("sgd", 1, 0), ("adam", 2, 64), ("adagrad", 3, 32), ("rowwise", 4, 4)):
        vdim = D + nstate
        t0 = rng.standard_normal((cap, vdim)).astype(np.float32)
        t0[:, D:] = 0
        table = T(t0).contiguous()
        idx = rng.choice(cap, size=N, replace=False).astype(np.int64)
        g = rng.standard_normal((N, D)).astype(np.float32)
        tptr = torch.tensor([table.data_ptr()], dtype=torch.int64, device=DEV)
        tv = torch.tensor([vdim], dtype=torch.int64, device=DEV)
        te = torch.tensor([D], dtype=torch.int64, device=DEV)
        tid = torch.zeros(N, dtype=torch.int64, device=DEV)
        rows = t0[idx].copy()
        if kind == 1:
            e.sgd_update_for_flat_table(T(g), T(idx), tptr, tid, tv, te, D, True, 0.1)
            orc.sgd_update(rows, g, D, 0.1)
        elif kind == 2:
            e.adam_update_for_flat_table(T(g), T(idx), tptr, tid, tv, te, 0.01, 0.9, 0.999, 1e-8, 0.0, 3, D, True,
                                         e.DynamicEmbDataType.Float32.value)
            orc.adam_update(rows, g, D, 0.01, 0.9, 0.999, 1e-8, 0.0, 3)
        elif kind == 3:
            e.adagrad_update_for_flat_table(T(g), T(idx), tptr, tid, tv, te, 0.1, 1e-8, D, True)
            orc.adagrad_update(rows, g, D, 0.1, 1e-8)
        else:
            e.rowwise_adagrad_for_flat_table(T(g), T(idx), tptr, tid, tv, te, 0.1, 1e-8, D, True)
            orc.rowwise_adagrad_update(rows, g, D, 0.1, 1e-8)
        np.testing.assert_allclose(table.cpu().numpy()[idx], rows, rtol=1e-5, atol=1e-6)
        # padded buffer variant: states start at max_emb_dim
        buf0 = rng.standard_normal((N, vdim)).astype(np.float32)
        buf0[:, D:] = 0
        buf = T(buf0).contiguous()
        rows = buf0.copy()
        if kind == 1:
            e.sgd_update_for_padded_buffer(T(g), buf, tid, te, D, vdim, True, 0.1); orc.sgd_update(rows, g, D, 0.1)
        elif kind == 2:
            e.adam_update_for_padded_buffer(T(g), buf, tid, te, D, vdim, True, 0.01, 0.9, 0.999, 1e-8, 0.0, 2)
            orc.adam_update(rows, g, D, 0.01, 0.9, 0.999, 1e-8, 0.0, 2)
        elif kind == 3:
            e.adagrad_update_for_padded_buffer(T(g), buf, tid, te, D, vdim, True, 0.1, 1e-8); orc.adagrad_update(rows, g, D, 0.1, 1e-8)
        else:
            e.rowwise_adagrad_for_padded_buffer(T(g), buf, tid, te, D, vdim, True, 0.1, 1e-8)
            orc.rowwise_adagrad_update(rows, g, D, 0.1, 1e-8)
        np.testing.assert_allclose(buf.cpu().numpy(), rows, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ export / count / score blocks / dedup lengths
def _filled_table(ns=1, caps=(2048, 1024), n_keys=1500, seed=3):
    rng = np.random.default_rng(seed)
    tb = orc.OracleTable(list(caps), 128, ns)
    T_ = len(caps)
    keys = rng.choice(2**40, size=n_keys, replace=False).astype(np.uint64)
    tids = rng.integers(0, T_, n_keys).astype(np.int64)
    scores = rng.integers(1, 1000, n_keys).astype(np.uint64)
    tb.insert(keys, tids, scores, orc.POLICY_ASSIGN)
    storage = torch.from_numpy(tb.storage.copy()).to(DEV)
    return tb, storage


@pytest.mark.parametrize("ns", [1, 2])
@pytest.mark.parametrize("threshold", [None, 500])
def test_table_export_batch_and_count(ns, threshold):
    e = ext()
    tb, storage = _filled_table(ns)
    C = 128
    keys_v, _, scores_v = tb._view()
    flat_k = keys_v.reshape(-1)
    flat_s = scores_v.reshape(-1, ns)
    total = flat_k.size
    valid = (flat_k & np.uint64(0xFFFFFFFFFFFFFFFC)) != np.uint64(0xFFFFFFFFFFFFFFFC)
    for (offset, batch, table_begin, sidx) in [(0, total, 0, 0), (128 * 3, 128 * 7 + 5, 128 * 2, ns - 1), (total - 64, 64, 0, 0)]:
        sel = np.zeros(total, bool)
        sel[offset:offset + batch] = True
        m = valid & sel
        if threshold is not None:
            m &= flat_s[:, sidx] >= np.uint64(threshold)
        exp_idx = np.nonzero(m)[0]
        cnt, k, s, i = e.table_export_batch(storage, C, batch, offset, torch.int64, threshold, table_begin, ns, sidx)
        c = int(cnt.item())
        assert c == exp_idx.size
        np.testing.assert_array_equal(i[:c].cpu().numpy(), exp_idx - table_begin)  # slot order
        np.testing.assert_array_equal(k[:c].cpu().numpy().view(np.uint64), flat_k[exp_idx])
        np.testing.assert_array_equal(s[:c].cpu().numpy().view(np.uint64), flat_s[exp_idx, sidx])
        if threshold is not None:
            n = e.table_count_matched(storage, torch.int64, C, threshold, offset, offset + batch, ns, sidx)
            assert int(n.item()) == exp_idx.size
    n_all = e.table_count_matched(storage, torch.int64, C, 0, -1, -1, ns, 0)
    assert int(n_all.item()) == int(valid.sum())
    with pytest.raises(Exception):
        e.table_export_batch(storage, C, 128, total - 64, torch.int64, None, 0, ns)


def test_table_score_blocks():
    e = ext()
    ns = 2
    tb, storage = _filled_table(ns, caps=(1024, 2048))
    C = 128
    _, _, scores_v = tb._view()
    rng = np.random.default_rng(9)
    bkt_begin = 1024 // C  # table 1
    n = 300
    slots = rng.integers(0, 2048, n).astype(np.int64)
    slots[::17] = -1
    got = e.table_gather_score_blocks(storage, C, ns, bkt_begin, T(slots)).cpu().numpy().view(np.uint64)
    flat = scores_v.reshape(-1, ns)
    exp = np.where((slots >= 0)[:, None], flat[np.maximum(slots, 0) + bkt_begin * C], 0)
    np.testing.assert_array_equal(got, exp)
    # scatter new values to unique slots, read back
    uslots = rng.choice(2048, size=200, replace=False).astype(np.int64)
    vals = rng.integers(0, 2**62, (200, ns)).astype(np.int64)
    e.table_scatter_score_blocks(storage, C, ns, bkt_begin, T(uslots), T(vals))
    back = e.table_gather_score_blocks(storage, C, ns, bkt_begin, T(uslots)).cpu().numpy()
    np.testing.assert_array_equal(back, vals)
    # copy table1[uslots] -> a fresh table with bucket capacity 64, slots permuted
    dst = torch.zeros(4096 // 64 * 64 * (9 + 8 * ns), dtype=torch.uint8, device=DEV)
    dslots = rng.choice(4096, size=200, replace=False).astype(np.int64)
    e.table_copy_score_blocks(storage, C, dst, 64, ns, bkt_begin, 0, T(uslots), T(dslots))
    back2 = e.table_gather_score_blocks(dst, 64, ns, 0, T(dslots)).cpu().numpy()
    np.testing.assert_array_equal(back2, vals)


@pytest.mark.parametrize("tof,B", [([0, 1], 4), ([0, 2, 3], 5), ([0, 1, 4, 6], 3)])
def test_compute_dedup_lengths_and_segmented_sum(tof, B):
    e = ext()
    rng = np.random.default_rng(len(tof) + B)
    T_ = len(tof) - 1
    F = tof[-1]
    nu = rng.integers(0, 50, T_)
    nu[0] = 0 if T_ > 1 else 7
    uoff = np.concatenate([[0], np.cumsum(nu)]).astype(np.int64)
    nl, no = e.compute_dedup_lengths_cuda(T(uoff), T(np.array(tof, np.int64)), T_, B, F * B)
    nl, no = nl.cpu().numpy(), no.cpu().numpy()
    # restatement of lookup_kernel.cuh:1049-1090
    exp_l, exp_o = [], []
    for i in range(F * B):
        f = i // B
        t = max(tt for tt in range(T_) if tof[tt] <= f)
        buckets = (tof[t + 1] - tof[t]) * B
        bid = i - tof[t] * B
        base, rem = divmod(int(nu[t]), buckets)
        exp_l.append(base + (1 if bid < rem else 0))
        exp_o.append(int(uoff[t]) + bid * base + min(bid, rem))
    np.testing.assert_array_equal(nl, exp_l)
    np.testing.assert_array_equal(no[:-1], exp_o)
    assert no[-1] == uoff[-1] and nl.sum() == uoff[-1]
    for t in range(T_):  # every table's keys are covered contiguously by its own bags
        assert nl[tof[t] * B:tof[t + 1] * B].sum() == nu[t]
    empty_l, empty_o = e.compute_dedup_lengths_cuda(T(uoff), T(np.array(tof, np.int64)), T_, B, 0)
    assert empty_l.numel() == 0 and empty_o.tolist() == [0]
    data = rng.integers(0, 128, 5000).astype(np.int32)
    offs = np.sort(np.concatenate([[0, 5000], rng.integers(0, 5000, 6)])).astype(np.int64)
    got = e.segmented_sum_cuda(T(data), T(offs)).cpu().numpy()
    np.testing.assert_array_equal(got, [int(data[a:b].sum()) for a, b in zip(offs[:-1], offs[1:])])


@pytest.mark.parametrize("n,T,zipf", [(1, 1, False), (5000, 1, True), (40000, 3, True), (3000, 2, False)])
def test_segmented_unique_csr_and_group_by(n, T, zipf):
    """counts / ranks emitted by the forward dedup, and the CSR built from them, against the plain counting sort"""
    e = ext()
    TT = globals()['T']
    rng = np.random.default_rng(n + T)
    keys = (rng.zipf(1.2, n) % 5000 if zipf else rng.integers(0, 2000, n)).astype(np.int64)
    cuts = np.sort(rng.integers(0, n + 1, T - 1)) if T > 1 else np.zeros(0, np.int64)
    seg = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    uk, rev, uoff, cnt, rank = e.segmented_unique_csr(TT(keys), TT(seg), len(seg) - 1)
    uk_o, rev_o, uoff_o, _ = orc.segmented_unique(keys, seg)
    nu = int(uoff_o[-1])
    rev_n, cnt_n, rank_n = rev.cpu().numpy(), cnt.cpu().numpy()[:nu], rank.cpu().numpy()
    np.testing.assert_array_equal(rev_n, rev_o)
    np.testing.assert_array_equal(uk.cpu().numpy()[:nu].view(np.uint64), uk_o)
    np.testing.assert_array_equal(cnt_n, np.bincount(rev_o, minlength=nu))
    order = np.lexsort((rank_n, rev_n))
    starts = np.concatenate([[0], np.cumsum(cnt_n)])
    # ranks of every unique row are a permutation of 0..cnt-1
    np.testing.assert_array_equal(rank_n[order], np.concatenate([np.arange(c) for c in cnt_n]) if nu else np.zeros(0))
    # CSR from counts/ranks == CSR from the counting sort, row by row as sets (sequence mode: src = key position)
    p1, c1 = e.group_by_unique_csr(cnt, rank, rev, n, nu_dev=uoff[-1:])
    p0, c0 = e.group_by_unique(rev, n, nu_dev=uoff[-1:])
    p1, c1, p0, c0 = (x.cpu().numpy() for x in (p1, c1, p0, c0))
    np.testing.assert_array_equal(p1[:nu + 1], starts)
    np.testing.assert_array_equal(p0[:nu + 1], starts)
    for u in range(0, nu, max(1, nu // 200)):
        a, b = starts[u], starts[u + 1]
        assert sorted(c1[a:b].tolist()) == sorted(c0[a:b].tolist()) == np.nonzero(rev_o == u)[0].tolist()
    # pooled mode: src = bag id
    B = max(1, n // 7)
    lens = rng.multinomial(n, np.ones(B) / B)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    p2, c2 = e.group_by_unique_csr(cnt, rank, rev, n, offsets=TT(offs), nu_dev=uoff[-1:])
    c2 = c2.cpu().numpy()
    bag_of = np.repeat(np.arange(B), lens)
    for u in range(0, nu, max(1, nu // 200)):
        a, b = starts[u], starts[u + 1]
        assert sorted(c2[a:b].tolist()) == sorted(bag_of[rev_o == u].tolist())


def test_vmm_tensors_grow_in_place():
    """VMMTensor / HostVMMTensor (vmm_tensor.cu:555-585): extend maps more memory behind the SAME base address (address
    space reserved once, hipMemMap / hipHostRegister per chunk); old contents stay, new memory reads as zero, kernels
    address both flavours"""
    e = ext()
    for cls in (e.VMMTensor, e.HostVMMTensor):
        t = cls(1000, torch.float32, 0, reserve_numel=64_000_000)
        d = t.data()
        assert d.numel() == 1000 and t.allocated_numel() >= 1000 and t.allocated_bytes() >= 4000
        assert d.is_cuda == (cls is e.VMMTensor)
        p0 = d.data_ptr()
        d.copy_(torch.arange(1000, dtype=torch.float32))
        for n in (5_000_000, 40_000_000):
            t.extend(n)
            d2 = t.data()
            assert d2.numel() == n == t.logical_numel() and d2.data_ptr() == p0 == t.data_ptr()
            assert torch.equal(d2[:1000].cpu(), torch.arange(1000, dtype=torch.float32))
            assert float(d2[1000:].abs().sum()) == 0.0
        # a kernel reads and writes through the raw address (the host flavour over the host link)
        rows = d2.view(-1, 8)
        idx = torch.tensor([0, 3, 4_999_999], dtype=torch.int64, device=DEV)
        out = torch.empty(3, 8, device=DEV)
        addr = torch.tensor([p0], dtype=torch.int64, device=DEV)
        a = e.row_addresses(idx, None, addr, torch.tensor([8], dtype=torch.int64, device=DEV), 4)
        from mi355_native import check, lib, ptr, stream
        check(lib().mi355_gather_rows(None, 0, ptr(a), 0, None, 3, None, 8, ptr(out), 8, 0, 1, stream()), "gather_rows")
        assert torch.equal(out.cpu(), rows[idx.cpu()].cpu())
        with pytest.raises(Exception):
            t.extend(65_000_000 + (1 << 28))      # beyond the reservation
        del d, d2, rows, t


def test_block_bucketize_many_bags_full_oracle():
    """W * F * B past the single-block scan (multi-block offsets path): whole output against the oracle"""
    e = ext()
    rng = np.random.default_rng(77)
    W, F, B = 8, 2, 6000   # 96 000 destination bags
    lens = rng.integers(0, 4, size=F * B)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, 1 << 40, size=int(offsets[-1])).astype(np.int64)
    blk = np.array([(1 << 40) // W + 1] * F, np.int64)
    gl, gi, _, _, gperm = e.block_bucketize_sparse_features(T(lens.astype(np.int64)), T(idx), False, True,
                                                             T(np.array([1] * F, np.int32)), T(blk), W)
    onl, ono, oni, operm = orc.block_bucketize(offsets, idx, W, B, blk, 1)
    np.testing.assert_array_equal(gl.cpu().numpy(), onl)
    np.testing.assert_array_equal(gi.cpu().numpy(), oni.view(np.int64))
    np.testing.assert_array_equal(gperm.cpu().numpy(), operm)


@pytest.mark.parametrize("W", [1, 3, 8, 11])
@pytest.mark.parametrize("dist", [0, 1, 2])
def test_block_bucketize_short_bag_kernels(W, dist):
    """>= 8192 bags take the 8-lanes-per-bag kernels: ranks beyond one lane group (W = 11), bags longer than a lane
    group, empty bags and a ragged last wave, whole output against the oracle"""
    e = ext()
    rng = np.random.default_rng(W * 10 + dist)
    F, B = 3, 2803   # 8409 bags
    lens = rng.integers(0, 21, size=F * B)
    lens[rng.integers(0, F * B, 500)] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, 1 << 30, size=int(offsets[-1])).astype(np.int64)
    blk = np.array([(1 << 30) // W + 1] * F, np.int64)
    gl, gi, _, _, gperm = e.block_bucketize_sparse_features(T(lens.astype(np.int64)), T(idx), False, True,
                                                             T(np.array([dist] * F, np.int32)), T(blk), W)
    onl, ono, oni, operm = orc.block_bucketize(offsets, idx, W, B, blk, dist)
    np.testing.assert_array_equal(gl.cpu().numpy(), onl)
    np.testing.assert_array_equal(gi.cpu().numpy(), oni.view(np.int64))
    np.testing.assert_array_equal(gperm.cpu().numpy(), operm)
