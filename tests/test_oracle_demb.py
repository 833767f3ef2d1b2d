"""CPU tests that pin the path-A oracle (oracle/demb_oracle.c + oracle/oracle.py)
against the reference's own fixtures and invariants:

* hash / digest / empty digest: golden vectors generated from the reference's
  Python copy of the hash (tests/golden/gen_demb_golden.py);
* segmented unique: the invariants of corelib/dynamicemb/test/test_unique_op.py:81-150;
* table insert / lookup / evict: the invariants of
  test/unit_tests/table_operation/test_table_operation.py (round trip, bucket
  rule, eviction takes the lowest score, LOCKED protection inside one call);
* 11-key fixture of test_batched_dynamic_embedding_tables_v2.py:1517-1522 with
  the DEBUG initializer closed form of test/unit_tests/debug.py:157-224.
"""
import json
import os
import sys

import numpy as np
import pytest

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_hash_golden():
    g = json.load(open(os.path.join(GOLD, "demb_hash_golden.json")))
    for r in g["rows"]:
        k = int(r["key"])
        assert orc.fmix64(k) == int(r["fmix64"])
        assert orc.digest(k) == r["digest"]
        assert orc.hash64(k) == int(r["fmix64"]) & 0x7FFFFFFFFFFFFFFF
    assert orc.empty_digest() == g["empty_digest"]


def test_reserved_keys_invalid():
    t = orc.OracleTable([256])
    bad = np.array([orc.EMPTY_KEY, orc.RECLAIM_KEY, orc.LOCKED_KEY, 0xFFFFFFFFFFFFFFFC], dtype=np.uint64)
    idx, res, _ = t.insert(bad, np.zeros(4, np.int64), np.ones(4, np.uint64))
    assert (idx == -1).all() and (res == orc.RES_ILLEGAL).all()
    _, f, i = t.lookup(bad, np.zeros(4, np.int64))
    assert not f.any() and (i == -1).all()


def _rand_keys(rng, n, hi=1 << 40):
    return rng.choice(hi, size=n, replace=False).astype(np.int64)


def test_unique_invariants():
    rng = np.random.default_rng(0)
    T = 3
    lens = [1000, 0, 2500]
    keys = np.concatenate([rng.integers(0, 300, size=n) for n in lens]).astype(np.int64)
    seg = np.concatenate([[0], np.cumsum(lens)])
    uk, oi, to, fr = orc.segmented_unique(keys, seg, count_freq=True)
    # test_unique_op.py:81-105
    assert (uk.view(np.int64)[oi] == keys).all()
    assert (np.diff(to) >= 0).all() and to[-1] == uk.size
    for t in range(T):
        a = keys[seg[t]:seg[t + 1]]
        assert to[t + 1] - to[t] == np.unique(a).size
        # first-occurrence order (this build's deterministic choice)
        _, first = np.unique(a, return_index=True)
        assert (uk.view(np.int64)[to[t]:to[t + 1]] == a[np.sort(first)]).all()
    assert fr.sum() == keys.size
    # same key in two tables -> two uniques (test_unique_op.py:110-150)
    uk2, _, to2, _ = orc.segmented_unique(np.array([5, 5, 5], np.int64), np.array([0, 2, 3]))
    assert uk2.size == 2 and list(to2) == [0, 1, 2]
    # weighted frequencies
    _, oi3, _, fr3 = orc.segmented_unique(np.array([7, 9, 7], np.int64), np.array([0, 3]),
                                          in_freq=np.array([2, 3, 4]), count_freq=True)
    assert list(fr3) == [6, 3] and list(oi3) == [0, 1, 0]


def test_insert_lookup_roundtrip_and_bucket_rule():
    rng = np.random.default_rng(1)
    C = 128
    t = orc.OracleTable([4096, 1024], bucket_capacity=C)
    keys = _rand_keys(rng, 3000)
    tids = (rng.random(3000) < 0.25).astype(np.int64)
    scores = rng.integers(1, 1 << 30, size=3000).astype(np.uint64)
    idx, res, so = t.insert(keys, tids, scores, orc.POLICY_ASSIGN)
    assert (res == orc.RES_INSERT).all() or ((res == orc.RES_INSERT) | (res == orc.RES_EVICT)).all()
    so2, found, idx2 = t.lookup(keys, tids)
    ok = idx >= 0
    assert found[ok].all() and (idx2[ok] == idx[ok]).all()
    assert (so2[ok].astype(np.uint64) == scores[ok]).all()
    # bucket rule kernels.cuh:107-125: slot // C == (hash % table_cap) // C
    for k, tid, i in zip(keys[:200], tids[:200], idx[:200]):
        cap = t.per_table_capacity[tid]
        assert i // C == (orc.hash64(int(k)) % cap) // C
    # key <-> slot bijection per table
    for tid in (0, 1):
        s = idx[(tids == tid) & ok]
        assert np.unique(s).size == s.size
    assert t.bucket_sizes.sum() == (res == orc.RES_INSERT).sum()
    # unknown keys are not found
    _, f3, i3 = t.lookup(keys + (1 << 41), tids)
    assert not f3.any() and (i3 == -1).all()


def test_eviction_takes_lowest_score_and_locks_protect_new_entries():
    C = 16
    t = orc.OracleTable([C], bucket_capacity=C)  # one bucket
    keys = np.arange(100, 100 + C, dtype=np.int64)
    scores = np.arange(10, 10 + C, dtype=np.uint64)
    idx, res, _ = t.insert(keys, np.zeros(C, np.int64), scores, orc.POLICY_ASSIGN)
    assert (res == orc.RES_INSERT).all() and sorted(idx) == list(range(C))
    # one more key evicts the min score (key 100, score 10)
    idx2, res2, _, ev = t.insert(np.array([999], np.int64), np.zeros(1, np.int64),
                                 np.array([50], np.uint64), orc.POLICY_ASSIGN, evict_out=True)
    assert res2[0] == orc.RES_EVICT and ev[0][0] == 100 and ev[2][0] == 10 and ev[1][0] == idx2[0]
    assert idx2[0] == idx[0]
    _, f, _ = t.lookup(np.array([100, 999], np.int64), np.zeros(2, np.int64))
    assert list(f) == [False, True]
    # a full batch of C+1 new keys in ONE call: C succeed by evicting, the last is BUSY
    newk = np.arange(5000, 5000 + C + 1, dtype=np.int64)
    idx3, res3, _, ev3 = t.insert(newk, np.zeros(C + 1, np.int64), np.full(C + 1, 7, np.uint64),
                                  orc.POLICY_ASSIGN, evict_out=True)
    assert (res3[:C] == orc.RES_EVICT).all() and res3[C] == orc.RES_BUSY and idx3[C] == -1
    assert ev3[1][-1] == -(C + 1) and ev3[0][-1] == newk[C]
    # pinned slots (ref counter > 0) are never evicted
    t.counter[:] = 1
    _, res4, _ = t.insert(np.array([31337], np.int64), np.zeros(1, np.int64), np.array([1], np.uint64),
                          orc.POLICY_ASSIGN)
    assert res4[0] == orc.RES_BUSY


def test_erase_and_reclaim():
    C = 16
    t = orc.OracleTable([C], bucket_capacity=C)
    keys = np.arange(1, C + 1, dtype=np.int64)
    idx, _, _ = t.insert(keys, np.zeros(C, np.int64), np.full(C, 5, np.uint64), orc.POLICY_ASSIGN)
    e = t.erase(keys[:3], np.zeros(3, np.int64))
    assert (e == idx[:3]).all() and t.bucket_sizes[0] == C - 3
    _, f, _ = t.lookup(keys, np.zeros(C, np.int64))
    assert list(f[:3]) == [False] * 3 and f[3:].all()
    # tombstone has score 0 -> reused first, lowest slot first (types.cuh:398-512)
    i2, r2, _ = t.insert(np.array([77], np.int64), np.zeros(1, np.int64), np.array([9], np.uint64),
                         orc.POLICY_ASSIGN)
    assert r2[0] == orc.RES_RECLAIM and i2[0] == min(idx[:3]) and t.bucket_sizes[0] == C - 2


def test_policies():
    t = orc.OracleTable([256])
    k = np.array([42], np.int64)
    z = np.zeros(1, np.int64)
    t.insert(k, z, np.array([5], np.uint64), orc.POLICY_ASSIGN)
    so, f, _ = t.lookup(k, z, np.array([3], np.uint64), orc.POLICY_ACCUMULATE)
    assert f[0] and so[0] == 8
    so, _, _ = t.lookup(k, z)  # CONST returns the stored score
    assert so[0] == 8
    so, _, _ = t.lookup(k, z, None, orc.POLICY_GLOBAL_TIMER, timer=123456)
    assert so[0] == 123456
    t2 = orc.OracleTable([256], num_scores=2)
    t2.insert(k, z, np.array([2], np.uint64), orc.POLICY_LRU_LFU, timer=1000)
    so, _, _ = t2.lookup(k, z, np.array([3], np.uint64), orc.POLICY_LRU_LFU, timer=2000)
    assert so[0] == 5
    _, _, sc = t2._view()
    assert 2000 in sc[..., 0] and 5 in sc[..., 1]


def test_deterministic_insert_is_order_independent():
    rng = np.random.default_rng(3)
    keys = _rand_keys(rng, 5000)
    tids = np.zeros(5000, np.int64)
    sc = rng.integers(1, 1000, size=5000).astype(np.uint64)
    a = orc.OracleTable([2048], bucket_capacity=128)
    b = orc.OracleTable([2048], bucket_capacity=128)
    ia = a.insert_deterministic(keys, tids, sc, orc.POLICY_ASSIGN)
    p = rng.permutation(5000)
    ib = b.insert_deterministic(keys[p], tids, sc[p], orc.POLICY_ASSIGN)
    assert (ia[p] == ib).all()
    assert (a.storage == b.storage).all()
    assert a.bucket_sizes.sum() == 2048  # overfull table: 5000 keys into 2048 slots


def test_eleven_key_fixture_debug_closed_form():
    """test_batched_dynamic_embedding_tables_v2.py:1517-1522 key stream; DEBUG initializer
    (row = key % 100000) => pooled SUM = sum of keys of the bag (debug.py:157-180)."""
    indices = np.array([0, 1, 12, 64, 8, 12, 15, 2, 7, 105, 0], np.int64)
    offsets = np.array([0, 2, 3, 5, 6, 8, 10, 10, 11], np.int64)
    feature_offsets = np.array([0, 2, 3, 4], np.int64)  # feature_table_map = [0,0,1,2]
    B, D = 2, 8
    rng_ = orc.get_table_range(offsets, feature_offsets, B)
    assert list(rng_) == [0, 6, 10, 11]
    uk, rev, to, _ = orc.segmented_unique(indices, rng_)
    assert list(to) == [0, 5, 9, 10]
    tids = orc.expand_table_ids(to, uk.size)
    assert list(tids) == [0] * 5 + [1] * 4 + [2]
    t = orc.OracleTable([2048] * 3)
    slots = t.insert_deterministic(uk, tids, np.ones(uk.size, np.uint64), orc.POLICY_ASSIGN)
    assert (slots >= 0).all()
    unique_embs = orc.debug_init(uk, D)
    out = orc.gather_pooled(unique_embs, rev, offsets, B, combiner=0)
    exp = np.zeros((B, 4 * D), np.float32)
    for i in range(8):
        f, b = divmod(i, B)
        exp[b, f * D:(f + 1) * D] = float(indices[offsets[i]:offsets[i + 1]].sum())
    assert (out == exp).all()
    assert (orc.gather_pooled_fast(unique_embs, rev, offsets, B, 0) == exp).all()
    mean = orc.gather_pooled(unique_embs, rev, offsets, B, combiner=1)
    assert mean[0, 0] == 0.5 and mean[1, 3 * D] == 0.0 and mean[0, 3 * D] == 0.0
    seq = orc.gather_sequence(unique_embs, rev)
    assert (seq[:, 0] == indices % 100000).all()


def test_reduce_grads_matches_dense_autograd():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    B, F, D, Nu = 6, 3, 8, 20
    lens = rng.integers(0, 4, size=F * B)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rev = rng.integers(0, Nu, size=int(offsets[-1])).astype(np.int64)
    w = torch.randn(Nu, D, dtype=torch.float64, requires_grad=True)
    for combiner in (0, 1):
        g = rng.standard_normal((B, F * D)).astype(np.float32)
        ug = orc.reduce_grads(rev, g, Nu, B, offsets, None, combiner)
        ugf = orc.reduce_grads_pooled_fast(rev, g, Nu, B, offsets, combiner)
        # dense autograd of the same pooled forward
        out = torch.zeros(B, F * D, dtype=torch.float64)
        rows = []
        for i in range(F * B):
            f, b = divmod(i, B)
            s = w[rev[offsets[i]:offsets[i + 1]]].sum(0)
            if combiner == 1 and lens[i] > 0:
                s = s / float(lens[i])
            rows.append((b, f, s))
        loss = sum((s * torch.from_numpy(g[b, f * D:(f + 1) * D].astype(np.float64))).sum() for b, f, s in rows)
        (gw,) = torch.autograd.grad(loss, w)
        np.testing.assert_allclose(ug, gw.numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ugf, gw.numpy(), rtol=1e-5, atol=1e-5)


def test_optimizers_match_torch_optim():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(7)
    N, D = 16, 8
    w0 = rng.standard_normal((N, D)).astype(np.float32)
    gs = [rng.standard_normal((N, D)).astype(np.float32) for _ in range(4)]

    def run_torch(opt_ctor):
        p = torch.nn.Parameter(torch.from_numpy(w0.copy()))
        o = opt_ctor([p])
        for g in gs:
            p.grad = torch.from_numpy(g.copy())
            o.step()
        return p.detach().numpy()

    rows = w0.copy()
    for g in gs:
        orc.sgd_update(rows, g, D, 0.3)
    np.testing.assert_allclose(rows, run_torch(lambda p: torch.optim.SGD(p, lr=0.3)), rtol=1e-6, atol=1e-6)

    rows = np.concatenate([w0.copy(), np.zeros((N, 2 * D), np.float32)], 1)
    for it, g in enumerate(gs, 1):
        orc.adam_update(rows, g, D, 0.01, 0.9, 0.999, 1e-8, 0.0, it)
    # the reference's Adam divides by (sqrt(vhat)+eps) like torch.optim.Adam
    np.testing.assert_allclose(rows[:, :D], run_torch(lambda p: torch.optim.Adam(p, lr=0.01, eps=1e-8)),
                               rtol=2e-5, atol=2e-6)

    rows = np.concatenate([w0.copy(), np.zeros((N, D), np.float32)], 1)
    for g in gs:
        orc.adagrad_update(rows, g, D, 0.1, 1e-10)
    np.testing.assert_allclose(rows[:, :D], run_torch(lambda p: torch.optim.Adagrad(p, lr=0.1, eps=1e-10)),
                               rtol=1e-5, atol=1e-6)

    rows = np.concatenate([w0.copy(), np.zeros((N, 4), np.float32)], 1)
    for g in gs:
        orc.rowwise_adagrad_update(rows, g, D, 0.1, 1e-8)
    G = np.zeros(N, np.float32)
    w = w0.copy()
    for g in gs:
        G += (g * g).mean(1)
        w -= 0.1 * g / (np.sqrt(G)[:, None] + 1e-8)
    np.testing.assert_allclose(rows[:, :D], w, rtol=1e-5, atol=1e-6)


def test_block_bucketize_routing():
    rng = np.random.default_rng(11)
    W, F, B = 4, 2, 3
    lens = rng.integers(0, 5, size=F * B)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, 1000, size=int(offsets[-1])).astype(np.int64)
    for dist in (1, 2, 0):
        nl, no, ni, perm = orc.block_bucketize(offsets, idx, W, B, [250, 250], dist)
        assert nl.sum() == idx.size and no[-1] == idx.size
        assert sorted(perm) == list(range(idx.size))
        for j, k in enumerate(idx):
            p = int(np.searchsorted(no, perm[j], side="right") - 1) // (F * B)
            if dist == 1:
                assert p == k % W and ni[perm[j]] == k
            elif dist == 2:
                assert p == orc.fmix64(int(k)) % W and ni[perm[j]] == k
            else:
                assert p == k // 250 and ni[perm[j]] == k % 250


def test_block_bucketize_variable_batch_and_uneven_boundaries_against_a_loop():
    """orc_block_bucketize_ex against a plain Python loop that follows the reference kernels line by line
    (sparse_block_bucketize_features.cu:240-292 lengths, :316-362 scatter)"""
    rng = np.random.default_rng(5)
    W = 4
    bs = [3, 0, 2]
    F, FB = len(bs), sum(bs)
    bag_feature = np.repeat(np.arange(F), bs)
    lens = rng.integers(0, 7, size=FB)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rng.integers(0, 1000, size=int(offsets[-1])).astype(np.int64)
    pos = [np.array([10, 200, 400, 800, 990], np.int64), np.array([0, 1, 2, 3, 4], np.int64), np.array([0, 100, 100, 500, 1000], np.int64)]
    blk = [250, 250, 250]
    for use_pos in (False, True):
        for dist in ([0, 0, 0], [1, 2, 0]):
            nl, no, ni, perm = orc.block_bucketize_ex(offsets, idx, W, 1, blk, dist, bag_feature, pos if use_pos else None)
            want = [[[] for _ in range(FB)] for _ in range(W)]
            for b in range(FB):
                f = bag_feature[b]
                for j in range(offsets[b], offsets[b + 1]):
                    k = int(idx[j])
                    if use_pos:
                        lb = int(np.searchsorted(pos[f], k, side="right")) - 1
                        p, nw = (lb, k - int(pos[f][lb])) if 0 <= lb < W else (k % W, k // W)
                    elif dist[f] == 1:
                        p, nw = k % W, k
                    elif dist[f] == 2:
                        p, nw = orc.fmix64(k) % W, k
                    else:
                        p, nw = (k // blk[f], k % blk[f]) if k < blk[f] * W else (k % W, k // W)
                    want[p][b].append((nw, j))
            flat = [x for p in range(W) for b in range(FB) for x in want[p][b]]
            assert [x[0] for x in flat] == ni.view(np.int64).tolist()
            assert [len(want[p][b]) for p in range(W) for b in range(FB)] == nl.tolist()
            for dst, (_, j) in enumerate(flat):
                assert perm[j] == dst


# ------------------------------------------------------------------------------------------------------------------------
# pins from the reference's own pure-Python code (tests/golden/gen_demb_flow_golden.py pulls it out of the AST)
FLOW = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demb_flow_golden.npz"))


@pytest.mark.parametrize("name", [str(x) for x in FLOW["flow_cases"]])
def test_deterministic_insert_follows_the_reference_wave_order(name):
    """oracle.insert_deterministic == the reference's `_bucketize_and_pad` + `_deterministic_insert`
    (scored_hashtable.py:1451-1558) driving the same oracle kernels: same slot for every key, same arena bytes"""
    g = lambda k: FLOW[f"{name}/{k}"]
    C, policy, nb = [int(x) for x in g("C")]
    t = orc.OracleTable([int(c) for c in g("caps")], C)
    for i in range(nb):
        idx = t.insert_deterministic(g(f"keys{i}"), g(f"tids{i}"), g(f"scores{i}"), policy)
        assert np.array_equal(idx, g(f"idx{i}")), f"batch {i}"
    assert np.array_equal(t.storage, g("arena")) and np.array_equal(t.bucket_sizes, g("bucket_sizes"))


def test_optimizer_state_layout_matches_the_reference_functions():
    """row = [embedding | optimizer state]: get_optimizer_state_dim / get_optimizer_ckpt_state_dim (optimizer.py:36-75)"""
    import torch

    from dynamicemb.dynamicemb_config import EmbOptimType, get_optimizer_state_dim
    from dynamicemb.optimizer import get_optimizer_ckpt_state_dim

    names = [str(x) for x in FLOW["opt_names"]]
    dts = [torch.float32, torch.bfloat16, torch.float16]
    for oi, dim, dc, state, ckpt in FLOW["opt_rows"]:
        o = EmbOptimType[names[oi]]
        assert get_optimizer_state_dim(o, int(dim), dts[dc]) == state, (o, dim, dts[dc])
        assert get_optimizer_ckpt_state_dim(o, int(dim)) == ckpt, (o, dim)


@pytest.mark.parametrize("name", [str(x) for x in FLOW["zipf_cases"]])
def test_zipf_key_stream_is_the_reference_generator(name):
    """bench.zipf_keys reproduces dataset_generator.zipf (benchmark/dataset_generator.py:75-103) bit for bit under the
    same torch seed (CPU generator)"""
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    lo, hi, n, seed = [int(x) for x in FLOW[f"zipf/{name}/meta"]]
    torch.manual_seed(seed)
    got = bench.zipf_keys(lo, hi, float(FLOW[f"zipf/{name}/alpha"]), n, torch.device("cpu")).numpy()
    assert np.array_equal(got, FLOW[f"zipf/{name}/samples"])


# ----------------------------------------------------------------------------- overflow region of cache tables
@pytest.mark.parametrize("nb,C", [([1], 128), ([1, 3], 128), ([1, 2, 4], 64)])
def test_overflow_with_counter_oracle(nb, C):
    """the scenario of the reference's test_overflow_with_counter (test/unit_tests/table_operation/test_table_operation.py
    :676-1063) on the oracle restatement: fill + pin the main tables, overflow insertion in ten rounds, lookups through
    both regions, counter release, eviction, and what it implies for sizes / counters"""
    caps = [n * C for n in nb]
    Tn, ocap = len(caps), 3 * C
    o = orc.OracleTable(caps, bucket_capacity=C, enable_overflow=True)
    assert o.counter.size == sum(caps) + ocap * Tn
    main = np.array(caps)
    key = 1
    fk, ft, fi = [], [], []
    for _ in range(100):                                           # phase 2
        if all(o.bucket_sizes[o.tbo[t]:o.tbo[t + 1]].sum() == caps[t] for t in range(Tn)):
            break
        per = sum(caps)
        bk = np.concatenate([np.arange(key + t * per, key + (t + 1) * per) for t in range(Tn)])
        bt = np.repeat(np.arange(Tn), per)
        key += per * Tn
        idx, res, _ = o.insert(bk, bt, np.full(bk.size, 100, np.uint64))
        ok = np.isin(res, (0, 1, 2))
        _, still, _ = o.lookup(bk[ok], bt[ok])                     # (not displaced by a later key of the same call)
        sel = np.nonzero(ok)[0][still]
        o.counter[o.counter_index(idx[sel], bt[sel])] += 1
        fk.append(bk[sel]); ft.append(bt[sel]); fi.append(idx[sel])
    fk, ft, fi = np.concatenate(fk), np.concatenate(ft), np.concatenate(fi)
    assert all(o.bucket_sizes[o.tbo[t]:o.tbo[t + 1]].sum() == caps[t] for t in range(Tn))
    _, f, li = o.lookup_ovf(fk, ft)
    assert f.all() and np.array_equal(li, fi) and (o.counter[o.counter_index(fi, ft)] >= 1).all()   # phase 3
    pools = [np.arange(key + t * ocap, key + (t + 1) * ocap) for t in range(Tn)]
    key += Tn * ocap
    per = ocap // 10
    ak, at, ai = [], [], []
    for r in range(10):                                            # phase 4
        bk = np.concatenate([p[r * per:(r + 1) * per] for p in pools])
        bt = np.repeat(np.arange(Tn), per)
        idx, res, so, ev = o.insert_ovf(bk, bt, np.ones(bk.size, np.uint64))
        assert (res == 0).all() and ev[0].size == 0 and (idx >= main[bt]).all() and (idx < main[bt] + ocap).all()
        o.counter[o.counter_index(idx, bt)] += 1
        ak.append(bk); at.append(bt); ai.append(idx)
    ak, at, ai = np.concatenate(ak), np.concatenate(at), np.concatenate(ai)
    assert (o.ovf_sizes == 10 * per).all()
    _, f, li = o.lookup_ovf(fk, ft)                                # phase 5
    assert f.all() and np.array_equal(li, fi)
    so, f, li = o.lookup_ovf(ak, at)
    assert f.all() and np.array_equal(li, ai) and (so == 1).all()
    _, f, li = o.lookup_ovf(np.arange(key, key + 50), np.zeros(50, np.int64))
    assert not f.any() and (li == -1).all()
    _, f, _ = o.lookup(ak, at)                                     # the main-only lookup does not see them
    assert not f.any()
    o.counter[o.counter_index(fi, ft)] -= 1                        # phase 6
    o.counter[o.counter_index(ai, at)] -= 1
    assert (o.counter == 0).all()
    ek = np.arange(key + 100, key + 132)
    idx, res, so, ev = o.insert_ovf(ek, np.zeros(32, np.int64), np.full(32, 200, np.uint64))
    assert (res == 3).all() and ev[0].size == 32 and (idx < caps[0]).all()      # main-table evictions again
    assert np.isin(ev[0].view(np.int64), fk[ft == 0]).all() and (ev[2] == 100).all()


def test_overflow_oracle_evicts_by_counter_and_reports_busy():
    """overflow victims are the entries with ref-counter 0 in probe order (kernels.cuh:779-791); with every entry pinned
    the key is refused and reported as (key, -(i+1)) (kernels.cuh:797-799, 470-472)"""
    C = 16
    o = orc.OracleTable([C], bucket_capacity=C, enable_overflow=True)
    idx, _, _ = o.insert(np.arange(1, C + 1), np.zeros(C, np.int64), np.full(C, 5, np.uint64))
    o.counter[o.counter_index(idx, np.zeros(C, np.int64))] += 1
    k = np.arange(100, 100 + 3 * C)
    idx, res, _, ev = o.insert_ovf(k, np.zeros(k.size, np.int64), np.ones(k.size, np.uint64))
    assert (res == 0).all() and np.unique(idx).size == 3 * C and idx.min() == C and idx.max() == 4 * C - 1
    pin = idx[::2]
    o.counter[o.counter_index(pin, np.zeros(pin.size, np.int64))] += 1
    k2 = np.arange(1000, 1000 + 3 * C)            # 24 unpinned victims for 48 keys
    idx2, res2, _, ev = o.insert_ovf(k2, np.zeros(k2.size, np.int64), np.full(k2.size, 2, np.uint64))
    assert (res2 == 3).sum() == 3 * C // 2 and (res2 == 5).sum() == 3 * C // 2
    ek, ei, es, et = ev
    assert ek.size == 3 * C
    took = res2 == 3
    assert np.array_equal(np.sort(ek[ei >= 0].view(np.int64)), np.sort(k[1::2]))      # the unpinned half left
    assert np.array_equal(np.sort(ei[ei >= 0]), np.sort(idx[1::2])) and np.array_equal(np.sort(idx2[took]), np.sort(idx[1::2]))
    assert np.array_equal(ek[ei < 0].view(np.int64), k2[~took]) and np.array_equal(-(ei[ei < 0] + 1), np.nonzero(~took)[0])
    _, f, li = o.lookup_ovf(k[::2], np.zeros(pin.size, np.int64))
    assert f.all() and np.array_equal(li, pin)
    # A resident overflow key that comes again is an Assign on its slot (ACCUMULATE adds to its score) PROVIDED no
    # unpinned entry sits in front of it in probe order -- the single-pass scan would take that one as a victim first and
    # duplicate the key (kernels.cuh:755-791; the cache only inserts keys its lookup missed, so this does not arise there).
    o.counter[o.counter_index(idx2[took], np.zeros(int(took.sum()), np.int64))] += 1
    idx3, res3, so3, ev3 = o.insert_ovf(k[::2], np.zeros(pin.size, np.int64), np.full(pin.size, 10, np.uint64),
                                        orc.POLICY_ACCUMULATE)
    assert (res3 == 2).all() and np.array_equal(idx3, pin) and (so3 == 11).all() and ev3[0].size == 0


def test_planner_capacity_arithmetic_matches_the_reference_functions():
    """align_to_table_size / _sharded_table_bucket_layout / get_sharded_table_capacity / get_constraint_capacity
    (dynamicemb_config.py:661-760, executed out of the reference's AST into the fixture): the per-rank capacities and
    bucket layouts the sharding planner writes into DynamicEmbTableOptions"""
    import warnings

    import torch

    import dynamicemb as de
    from dynamicemb import dynamicemb_config as dc
    from dynamicemb.dynamicemb_config import EmbOptimType

    a, b, m = [int(x) for x in FLOW["plan_consts"]]
    assert (de.DEMB_TABLE_ALIGN_SIZE, de.BUCKET_ALIGNMENT, de.MAX_BUCKET_CAPACITY) == (a, b, m)
    for n, al, exp in FLOW["plan_align"]:
        assert dc.align_to_table_size(int(n), int(al)) == int(exp), (n, al)

    class Cfg:
        def __init__(self, n):
            self.num_embeddings = n

    for n, w, bc, nb, eff, cap in FLOW["plan_layout"]:
        assert dc._sharded_table_bucket_layout(Cfg(int(n)), int(w), int(bc)) == (int(nb), int(eff)), (n, w, bc)
        assert dc.get_sharded_table_capacity(Cfg(int(n)), int(w), int(bc)) == int(cap), (n, w, bc)
    names = [str(x) for x in FLOW["opt_names"]]
    dts = [torch.float32, torch.bfloat16, torch.float16]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for mem, dci, dim, oi, bc, cap in FLOW["plan_capacity"]:
            got = dc.get_constraint_capacity(int(mem), dts[dci], int(dim), EmbOptimType[names[oi]], int(bc))
            assert got == int(cap), (mem, dts[dci], dim, names[oi], bc)


@pytest.mark.parametrize("W", [2, 3, 8])
@pytest.mark.parametrize("dist", ["continuous", "roundrobin", "hash_roundrobin"])
def test_routing_owner_matches_the_reference_cpu_function(W, dist):
    """owner rank of every key under the three routing rules = the reference's own `assign_owner_cpu`
    (test/unit_tests/test_hash_roundrobin_kuairand.py:27-45, executed out of its AST into the fixture): the oracle's
    block_bucketize must place every key in that owner's block"""
    from dynamicemb.input_dist import DIST_TYPES

    keys = FLOW["route/keys"]
    owner = FLOW[f"route/{dist}/{W}"]
    blk = (10_000_000 + W - 1) // W
    off = np.array([0, keys.size], np.int64)                       # one feature, one bag
    nl, no, ni, perm = orc.block_bucketize(off, keys, W, 1, np.array([blk], np.int64), DIST_TYPES[dist])
    assert np.array_equal(nl, np.bincount(owner, minlength=W))
    got_owner = np.empty(keys.size, np.int64)
    for p in range(W):
        got_owner[perm_positions(perm, no, p)] = p
    assert np.array_equal(got_owner, owner)


def perm_positions(perm, new_offsets, p):
    """input positions of the keys that landed in peer p's block (perm[i] = output position of input key i)"""
    lo, hi = int(new_offsets[p]), int(new_offsets[p + 1])
    return np.nonzero((perm >= lo) & (perm < hi))[0]
