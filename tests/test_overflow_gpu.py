"""Overflow buckets of cache tables (reference scored_hashtable.py:426-474,736-824, kernels.cuh:153-183,468-566,711-800):
the HIP kernels through the C ABI / the LinearBucketTable mirror against the sequential oracle restatement.

Bit-exact where the reference's algorithm is schedule independent (one key per call; single-bucket tables in
DEMB_DETERMINISM_MODE), through the invariants of the reference's own test (test_table_operation.py:676-1063) otherwise:
positions inside the overflow bucket depend on which of two colliding probes wins, on the GPU as in the reference."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"
INSERT, RECLAIM, ASSIGN, EVICT, BUSY = 0, 1, 2, 3, 5


def ext():
    import dynamicemb_extensions as e

    return e


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make(caps, C, policy=None, ns=1):
    from dynamicemb.scored_hashtable import ScoreSpec, get_scored_table

    e = ext()
    pol = policy if policy is not None else e.ScorePolicy.ASSIGN
    g = get_scored_table(capacity=list(caps), bucket_capacity=C, key_type=torch.int64, score_specs=[ScoreSpec("s", pol)],
                         device=torch.device(DEV), enable_overflow=True)
    o = orc.OracleTable(list(caps), bucket_capacity=C, num_scores=ns, enable_overflow=True)
    return g, o


def arg(values, policy=None):
    from dynamicemb.scored_hashtable import ScoreArg

    v = None if values is None else T(np.asarray(values, np.int64)).view(torch.uint64)
    return ScoreArg("s", v, policy)


def same_state(g, o):
    assert np.array_equal(g.table_storage_.cpu().numpy(), o.storage)
    assert np.array_equal(g.overflow_table_storage_.cpu().numpy(), o.ovf_storage)
    assert np.array_equal(g.bucket_sizes.cpu().numpy(), o.bucket_sizes)
    assert np.array_equal(g.overflow_bucket_sizes.cpu().numpy(), o.ovf_sizes)
    assert np.array_equal(g._ref_counter.cpu().numpy(), o.counter)


def fill_and_pin(g, o, caps, C, monkeypatch, policy=orc.POLICY_ASSIGN):
    """phase 2 of the reference test: fill the main tables (deterministic waves: bit-exact against the oracle) and pin
    every slot.  Returns (keys, tids, indices) of the resident keys and the next unused key."""
    monkeypatch.setenv("DEMB_DETERMINISM_MODE", "ON")
    key = 1
    ks, ts, xs = [], [], []
    for _ in range(100):
        if all(int(g.size(table_id=t) - g.overflow_bucket_sizes[t]) == caps[t] for t in range(len(caps))):
            break
        per = sum(caps)
        bk = np.concatenate([np.arange(key + t * per, key + (t + 1) * per) for t in range(len(caps))]).astype(np.int64)
        bt = np.repeat(np.arange(len(caps)), per).astype(np.int64)
        key += per * len(caps)
        res = torch.empty(bk.size, dtype=torch.uint8, device=DEV)
        idx = g.insert(T(bk), T(bt), arg(np.full(bk.size, 100), ext().ScorePolicy(policy)), res)
        io = o.insert_deterministic(bk, bt, np.full(bk.size, 100, np.uint64), policy)
        assert np.array_equal(idx.cpu().numpy(), io)
        ok = io >= 0
        # (deterministic mode reports final indices; keys evicted by a later wave of the same call come back as -1)
        g.increment_counter(T(io[ok]), T(bt[ok]))
        o.counter[o.counter_index(io[ok], bt[ok])] += 1
        ks.append(bk[ok]); ts.append(bt[ok]); xs.append(io[ok])
    monkeypatch.delenv("DEMB_DETERMINISM_MODE")
    for t in range(len(caps)):
        assert int(g.size(table_id=t)) == caps[t]
    same_state(g, o)
    return np.concatenate(ks), np.concatenate(ts), np.concatenate(xs), key


CONFIGS = [pytest.param([1], 128, id="1table_1bkt_cap128"), pytest.param([1, 3], 128, id="2tables_asym_cap128"),
           pytest.param([1, 2, 4], 64, id="3tables_mixed_cap64"), pytest.param([2, 1], 16, id="2tables_cap16")]


@pytest.mark.parametrize("nb,C", CONFIGS)
def test_construction(nb, C):
    """phase 1 of the reference test: capacities and counter layout"""
    caps = [n * C for n in nb]
    g, o = make(caps, C)
    assert g.enable_overflow_ is True and g.overflow_bucket_capacity_ == 3 * C
    total = sum(caps) + 3 * C * len(caps)
    assert g.capacity() == total and g._ref_counter.numel() == total and int(g._ref_counter.abs().sum()) == 0
    for t in range(len(caps)):
        assert g.main_capacity(table_id=t) == caps[t] and g.capacity(table_id=t) == caps[t] + 3 * C
    same_state(g, o)


@pytest.mark.parametrize("nb,C", CONFIGS)
def test_one_key_per_call_is_bit_exact(nb, C, monkeypatch):
    """One key per call has no schedule: indices, results, scores, evicted records and every byte of both arenas, the
    sizes and the counters equal the oracle's, through pinned-full main buckets, overflow inserts, overflow re-inserts
    (Assign), overflow evictions after the pins are released, and a full overflow bucket (Busy)."""
    e = ext()
    caps = [n * C for n in nb]
    g, o = make(caps, C, policy=e.ScorePolicy.ACCUMULATE)
    _, _, _, key = fill_and_pin(g, o, caps, C, monkeypatch, orc.POLICY_ACCUMULATE)
    rng = np.random.default_rng(C + len(nb))
    Tn = len(caps)
    pinned = []
    steps = 3 * C + 40   # past the overflow capacity of table 0
    for s in range(steps):
        t = 0 if s >= 2 * C else int(rng.integers(0, Tn))
        reinsert = s % 7 == 3 and pinned
        k = pinned[int(rng.integers(0, len(pinned)))][0] if reinsert else key + s
        if reinsert:
            t = [p[1] for p in pinned if p[0] == k][0]
        sc = int(rng.integers(1, 1000))
        res = torch.empty(1, dtype=torch.uint8, device=DEV)
        so = torch.empty(1, dtype=torch.int64, device=DEV)
        idx, h, ek, ei, es, et = g.insert_and_evict_with_counter_and_overflow(T(np.array([k])), T(np.array([t])), arg([sc]), res, so)
        io, ro, soo, (oek, oei, oes, oet) = o.insert_ovf(np.array([k]), np.array([t]), np.array([sc], np.uint64),
                                                         orc.POLICY_ACCUMULATE)
        assert int(idx[0]) == int(io[0]) and int(res[0]) == int(ro[0]) and int(so[0]) == int(soo[0]), f"step {s}"
        assert h == oek.size
        assert np.array_equal(ek.cpu().numpy().view(np.uint64), oek) and np.array_equal(ei.cpu().numpy(), oei)
        assert np.array_equal(es.cpu().numpy(), oes) and np.array_equal(et.cpu().numpy(), oet)
        if int(ro[0]) in (INSERT, EVICT) and s % 3 != 0:   # pin two thirds of the new entries
            g.increment_counter(idx, T(np.array([t])))
            o.counter[o.counter_index(io, np.array([t]))] += 1
            pinned.append((k, t, int(io[0])))
        if s == 2 * C:   # release some pins: evictions by counter == 0 become possible
            rel = pinned[::2]
            pinned = pinned[1::2]
            if rel:
                ri, rt = np.array([p[2] for p in rel]), np.array([p[1] for p in rel])
                g.decrement_counter(T(ri), T(rt))
                o.counter[o.counter_index(ri, rt)] -= 1
    same_state(g, o)
    assert int(g.overflow_bucket_sizes.sum()) > 0
    # pin the whole overflow bucket of table 0: it fills up, then refuses (Busy) -- still key by key against the oracle
    lo = sum(caps)
    g._ref_counter[lo:lo + 3 * C] += 1
    o.counter[lo:lo + 3 * C] += 1
    busy = 0
    for s in range(3 * C + 8):
        k = key + steps + s
        res = torch.empty(1, dtype=torch.uint8, device=DEV)
        idx, h, ek, ei, es, et = g.insert_and_evict_with_counter_and_overflow(T(np.array([k])), T(np.zeros(1, np.int64)), arg([1]), res)
        io, ro, _, (oek, oei, oes, oet) = o.insert_ovf(np.array([k]), np.zeros(1, np.int64), np.array([1], np.uint64),
                                                       orc.POLICY_ACCUMULATE)
        assert int(idx[0]) == int(io[0]) and int(res[0]) == int(ro[0])
        assert np.array_equal(ek.cpu().numpy().view(np.uint64), oek) and np.array_equal(ei.cpu().numpy(), oei)
        busy += int(ro[0]) == BUSY
        if busy >= 5:
            break
    assert busy >= 5
    same_state(g, o)
    steps += 3 * C + 8
    # lookups see the same thing (CONST and a score-updating policy)
    allk = np.arange(1, key + steps).astype(np.int64)
    for t in range(Tn):
        tt = np.full(allk.size, t, np.int64)
        so_g, f_g, i_g = g.lookup_with_overflow(T(allk), T(tt), arg(None, e.ScorePolicy.CONST))
        so_o, f_o, i_o = o.lookup_ovf(allk, tt)
        assert np.array_equal(f_g.cpu().numpy(), f_o) and np.array_equal(i_g.cpu().numpy(), i_o)
        assert np.array_equal(so_g.cpu().numpy(), so_o)
    tt = np.zeros(allk.size, np.int64)
    add = rng.integers(1, 50, allk.size).astype(np.int64)
    so_g, f_g, i_g = g.lookup_with_overflow(T(allk), T(tt), arg(add, e.ScorePolicy.ACCUMULATE))
    so_o, f_o, i_o = o.lookup_ovf(allk, tt, add.view(np.uint64), orc.POLICY_ACCUMULATE)
    assert np.array_equal(so_g.cpu().numpy(), so_o) and np.array_equal(i_g.cpu().numpy(), i_o)
    same_state(g, o)


def test_single_bucket_tables_deterministic_mode_is_bit_exact(monkeypatch):
    """tables of ONE bucket: a determinism-mode wave holds one key per table, so whole batches are schedule independent
    (the reference's _deterministic_insert_and_evict_with_overflow, scored_hashtable.py:1643-1735)"""
    C, caps = 32, [32, 32, 32]
    g, o = make(caps, C)
    _, _, _, key = fill_and_pin(g, o, caps, C, monkeypatch)
    monkeypatch.setenv("DEMB_DETERMINISM_MODE", "ON")
    rng = np.random.default_rng(3)
    for r in range(4):
        n = 60
        bk = np.arange(key, key + n).astype(np.int64)
        key += n
        bt = rng.integers(0, 3, n).astype(np.int64)
        sc = rng.integers(1, 1000, n).astype(np.int64)
        idx, h, ek, ei, es, et = g.insert_and_evict_with_counter_and_overflow(T(bk), T(bt), arg(sc))
        # oracle: the same waves (i-th key of every bucket in (bucket, key) order), then one CONST lookup
        ko, off, inv = o.bucketize(bk, bt)
        lens = np.diff(off)
        evs = [[], [], [], []]
        for w in range(int(lens.max())):
            sel = off[:-1][lens > w] + w
            _, _, _, ev = o.insert_ovf(ko[sel], bt[inv[sel]], sc.view(np.uint64)[inv[sel]], orc.POLICY_ASSIGN)
            for lst, a in zip(evs, ev):
                lst.append(a)
        _, _, io = o.lookup_ovf(bk, bt)
        assert np.array_equal(idx.cpu().numpy(), io)
        cat = [np.concatenate(x) for x in evs]
        assert h == cat[0].size
        order_g = np.lexsort((ek.cpu().numpy(), et.cpu().numpy()))
        order_o = np.lexsort((cat[0].view(np.int64), cat[3]))
        assert np.array_equal(ek.cpu().numpy()[order_g], cat[0].view(np.int64)[order_o])
        assert np.array_equal(et.cpu().numpy()[order_g], cat[3][order_o])
        assert np.array_equal(es.cpu().numpy()[order_g], cat[2][order_o])
        eg, eo = ei.cpu().numpy()[order_g], cat[1][order_o]
        assert np.array_equal(eg >= 0, eo >= 0) and np.array_equal(eg[eg >= 0], eo[eo >= 0])   # (< 0: wave-relative -(i+1))
        same_state(g, o)
        if r == 1:   # pin what sits in the overflow now, so later rounds run into Busy / cannot evict these
            ok = io >= 0
            g.increment_counter(T(io[ok]), T(bt[ok]))
            o.counter[o.counter_index(io[ok], bt[ok])] += 1


@pytest.mark.parametrize("nb,C", CONFIGS[:3])
def test_overflow_with_counter_flow(nb, C, monkeypatch):
    """the reference's own scenario (test_table_operation.py:676-1063, phases 2-7) with whole batches, checked through its
    invariants and, where the outcome is schedule independent, against the oracle: how many keys each table's overflow
    bucket takes, that nothing pinned is ever displaced, counters, reset."""
    e = ext()
    caps = [n * C for n in nb]
    Tn = len(caps)
    ocap = 3 * C
    g, o = make(caps, C)
    fk, ft, fi, key = fill_and_pin(g, o, caps, C, monkeypatch)
    main_caps = np.array(caps)
    # phase 4: ten rounds of overflow insertion; every success is pinned
    rounds, per = 10, ocap // 10
    pools = [np.arange(key + t * ocap, key + (t + 1) * ocap).astype(np.int64) for t in range(Tn)]
    key += Tn * ocap
    all_k, all_t, all_i, all_r = [], [], [], []
    for r in range(rounds):
        bk = np.concatenate([p[r * per:(r + 1) * per] for p in pools])
        bt = np.repeat(np.arange(Tn), per).astype(np.int64)
        res = torch.empty(bk.size, dtype=torch.uint8, device=DEV)
        idx, h, ek, ei, es, et = g.insert_and_evict_with_counter_and_overflow(T(bk), T(bt), arg(np.ones(bk.size)), res)
        io, ro, _, oev = o.insert_ovf(bk, bt, np.ones(bk.size, np.uint64), orc.POLICY_ASSIGN)
        rg, ig = res.cpu().numpy(), idx.cpu().numpy()
        # every main bucket is pinned and the overflow has room: all keys land in the overflow as fresh inserts -- the same
        # outcome (not the same positions) as the sequential oracle
        assert (rg == INSERT).all() and (ro == INSERT).all() and h == 0 and oev[0].size == 0
        assert (ig >= main_caps[bt]).all() and (ig < main_caps[bt] + ocap).all()
        for t in range(Tn):
            assert np.unique(ig[bt == t]).size == per            # distinct slots
        g.increment_counter(idx, T(bt))
        o.counter[o.counter_index(io, bt)] += 1
        all_k.append(bk); all_t.append(bt); all_i.append(ig); all_r.append(rg)
        assert np.array_equal(g.overflow_bucket_sizes.cpu().numpy(), o.ovf_sizes)
        assert np.array_equal(np.sort(g._ref_counter.cpu().numpy()), np.sort(o.counter))
    ak, at, ai = np.concatenate(all_k), np.concatenate(all_t), np.concatenate(all_i)
    # phase 5: lookups -- residents keep their main slots, overflow keys report the index their insert returned
    _, ff, fli = g.lookup_with_overflow(T(fk), T(ft), arg(None, e.ScorePolicy.CONST))
    assert bool(ff.all()) and np.array_equal(fli.cpu().numpy(), fi)
    so, of, oli = g.lookup_with_overflow(T(ak), T(at), arg(None, e.ScorePolicy.CONST))
    assert bool(of.all()) and np.array_equal(oli.cpu().numpy(), ai) and bool((so == 1).all())
    ghosts = np.arange(key, key + 50).astype(np.int64)
    _, gf, gi = g.lookup_with_overflow(T(ghosts), T(np.zeros(50, np.int64)), arg(None, e.ScorePolicy.CONST))
    assert not bool(gf.any()) and bool((gi == -1).all())
    # re-inserting resident overflow keys is an Assign on the same index with the new score
    res = torch.empty(ak.size, dtype=torch.uint8, device=DEV)
    idx, h, *_ = g.insert_and_evict_with_counter_and_overflow(T(ak), T(at), arg(np.full(ak.size, 9)), res)
    assert h == 0 and bool((res == ASSIGN).all()) and np.array_equal(idx.cpu().numpy(), ai)
    so, _, _ = g.lookup_with_overflow(T(ak), T(at), arg(None, e.ScorePolicy.CONST))
    assert bool((so == 9).all())
    # the overflow of every table holds 10*per pinned keys; fill the remainder and go past it: the surplus is refused (Busy)
    left = ocap - rounds * per
    extra = np.arange(key + 100, key + 100 + Tn * (left + 5)).astype(np.int64)
    et_ = np.repeat(np.arange(Tn), left + 5).astype(np.int64)
    res = torch.empty(extra.size, dtype=torch.uint8, device=DEV)
    idx, h, ek, ei, es, ett = g.insert_and_evict_with_counter_and_overflow(T(extra), T(et_), arg(np.full(extra.size, 5)), res)
    rg = res.cpu().numpy()
    for t in range(Tn):
        assert int((rg[et_ == t] == INSERT).sum()) == left and int((rg[et_ == t] == BUSY).sum()) == 5
    assert h == 5 * Tn and bool((ei < 0).all())
    refused = extra[rg == BUSY]
    assert np.array_equal(np.sort(ek.cpu().numpy()), np.sort(refused))                # a refused key reports itself ...
    assert np.array_equal(np.sort(-(ei.cpu().numpy() + 1)), np.sort(np.nonzero(rg == BUSY)[0]))   # ... at -(i+1)
    assert bool((idx[T(rg == BUSY)] == -1).all())
    unp_k, unp_t, unp_i = extra[rg == INSERT], et_[rg == INSERT], idx.cpu().numpy()[rg == INSERT]
    # phase 6: victims in the overflow are exactly the unpinned entries; pinned ones survive
    newk = np.arange(key + 10_000, key + 10_000 + Tn * left).astype(np.int64)
    nt = np.repeat(np.arange(Tn), left).astype(np.int64)
    res = torch.empty(newk.size, dtype=torch.uint8, device=DEV)
    idx, h, ek, ei, es, ett = g.insert_and_evict_with_counter_and_overflow(T(newk), T(nt), arg(np.full(newk.size, 7)), res)
    assert bool((res == EVICT).all()) and h == newk.size
    assert np.array_equal(np.sort(ek.cpu().numpy()), np.sort(unp_k))
    assert np.array_equal(np.sort(ei.cpu().numpy() + 10**9 * ett.cpu().numpy()), np.sort(unp_i + 10**9 * unp_t))
    assert np.array_equal(np.sort(idx.cpu().numpy() + 10**9 * nt), np.sort(unp_i + 10**9 * unp_t))
    _, of, oli = g.lookup_with_overflow(T(ak), T(at), arg(None, e.ScorePolicy.CONST))
    assert bool(of.all()) and np.array_equal(oli.cpu().numpy(), ai)
    _, uf, _ = g.lookup_with_overflow(T(unp_k), T(unp_t), arg(None, e.ScorePolicy.CONST))
    assert not bool(uf.any())
    # release every pin: counters return to zero, main-table eviction works again
    g.decrement_counter(T(fi), T(ft))
    g.decrement_counter(T(ai), T(at))
    assert int(g._ref_counter.abs().sum()) == 0
    ev = np.arange(key + 20_000, key + 20_032).astype(np.int64)
    res = torch.empty(32, dtype=torch.uint8, device=DEV)
    idx, h, ek, *_ = g.insert_and_evict_with_counter_and_overflow(T(ev), T(np.zeros(32, np.int64)), arg(np.full(32, 200)), res)
    assert h > 0 and bool((res == EVICT).any()) and bool((idx[res == EVICT] < caps[0]).all())
    # phase 7
    g.reset()
    for t in range(Tn):
        assert int(g.size(table_id=t)) == 0
    assert int(g._ref_counter.abs().sum()) == 0 and int(g.overflow_bucket_sizes.sum()) == 0
    g2, o2 = make(caps, C)
    assert np.array_equal(g.overflow_table_storage_.cpu().numpy(), o2.ovf_storage)


def test_overflow_lru_lfu_score_blocks(monkeypatch):
    """two score words: a fresh overflow slot restarts its frequency at the inserted value and stamps the timer, a hit
    accumulates (kernels.cuh:494-513); one key per call against the oracle"""
    e = ext()
    C, caps = 16, [16]
    g, o = make(caps, C, policy=e.ScorePolicy.LRU_LFU, ns=2)
    monkeypatch.setattr(e, "TIMER_OVERRIDE", 5000)
    _, _, _, key = fill_and_pin_lfu(g, o, caps, C)
    for s in range(30):
        k = key + (s % 11)
        monkeypatch.setattr(e, "TIMER_OVERRIDE", 6000 + s)
        idx, h, *_ = g.insert_and_evict_with_counter_and_overflow(T(np.array([k])), T(np.zeros(1, np.int64)), arg([s + 1]))
        io, ro, soo, _ = o.insert_ovf(np.array([k]), np.zeros(1, np.int64), np.array([s + 1], np.uint64), orc.POLICY_LRU_LFU,
                                      timer=6000 + s)
        assert int(idx[0]) == int(io[0])
        so_g, f_g, i_g = g.lookup_with_overflow(T(np.array([k])), T(np.zeros(1, np.int64)), arg([2], e.ScorePolicy.LRU_LFU))
        so_o, f_o, i_o = o.lookup_ovf(np.array([k]), np.zeros(1, np.int64), np.array([2], np.uint64), orc.POLICY_LRU_LFU,
                                      timer=6000 + s)
        assert int(so_g[0]) == int(so_o[0]) and bool(f_g[0]) == bool(f_o[0])
    same_state(g, o)


def fill_and_pin_lfu(g, o, caps, C):
    """one key per call fill (schedule independent for any policy) + pin"""
    e = ext()
    key = 1
    ks, xs = [], []
    for i in range(caps[0]):
        idx = g.insert(T(np.array([key + i])), T(np.zeros(1, np.int64)), arg([3]))
        io, _, _ = o.insert(np.array([key + i]), np.zeros(1, np.int64), np.array([3], np.uint64), orc.POLICY_LRU_LFU,
                            timer=e.TIMER_OVERRIDE)
        assert int(idx[0]) == int(io[0])
        ks.append(key + i); xs.append(int(io[0]))
    xs = np.array(xs)
    g.increment_counter(T(xs), T(np.zeros(xs.size, np.int64)))
    o.counter[o.counter_index(xs, np.zeros(xs.size, np.int64))] += 1
    same_state(g, o)
    return np.array(ks), None, xs, key + caps[0]
