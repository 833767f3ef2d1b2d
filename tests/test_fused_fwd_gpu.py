"""GPU tests of the fused index stage (csrc/fused_fwd.hip, mi355_demb_forward_fused) against the per-op chain
(MI355_FUSED=0: segmented_unique + table_lookup + table_insert + unlock/init) and against the oracle's dict twin:
same pooled / sequence outputs bit for bit, same rows after training steps, same table contents; dedup of one new key
arriving from many tiles; hits + misses in a full bucket (a key the batch just hit must never be evicted by the batch's
own inserts: the reference pins found slots before the insert, _prefetch_hbm_direct_path
batched_dynamicemb_function.py:559-696)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(fused, dims=(16,), cap=4096, pooling="SUM", opt="SGD", strategy="TIMESTAMP", bucket=128, fmap=None, monkeypatch=None, **kw):
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2 as B2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs as IA, DynamicEmbInitializerMode as IM,
                                              DynamicEmbPoolingMode as PM, DynamicEmbScoreStrategy as SS,
                                              DynamicEmbTableOptions as TO, EmbOptimType as OT)
    monkeypatch.setenv("MI355_FUSED", "1" if fused else "0")
    opts = [TO(dim=d, max_capacity=cap, index_type=torch.int64, embedding_dtype=torch.float32, bucket_capacity=bucket,
               initializer_args=IA(mode=IM.UNIFORM, lower=-0.5, upper=0.5), score_strategy=getattr(SS, strategy)) for d in dims]
    m = B2(table_options=opts, feature_table_map=fmap or list(range(len(dims))), pooling_mode=getattr(PM, pooling),
           optimizer=getattr(OT, opt), output_dtype=torch.float32, device=torch.device(DEV), **kw)
    assert m._fused == bool(fused)
    return m


def _counters_clear(m):
    """header + per-slot occurrence counters of the fused forward are all zero between steps (the unique-id map behind
    them is scratch)"""
    torch.cuda.synchronize()
    cap = m.table.capacity_   # aux = [hdr 64][partition counters 4 x 4096][{occ, uid} x (S + 1)][locks]
    H = 64 + 4 * 4096         # header, partition counters and the occ halves must be zero between steps
    return int(m._fused_aux[:H].abs().sum()) == 0 and int(m._fused_aux[H: H + 2 * (cap + 1): 2].abs().sum()) == 0


def _batch(rng, F, B, hi, maxlen=6):
    lens = rng.integers(0, maxlen, size=F * B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    keys = rng.integers(0, hi, size=int(lens.sum())).astype(np.int64)
    return torch.from_numpy(keys).to(DEV), torch.from_numpy(off).to(DEV)


@pytest.mark.parametrize("pooling", ["SUM", "MEAN", "NONE"])
@pytest.mark.parametrize("opt", ["SGD", "ADAM", "EXACT_ROWWISE_ADAGRAD"])
@pytest.mark.parametrize("strategy", ["TIMESTAMP", "STEP", "LFU"])
def test_fused_matches_per_op_chain_over_training_steps(pooling, opt, strategy, monkeypatch):
    dims = (16, 16, 16) if pooling == "NONE" else (8, 16, 32)
    ref = _mk(False, dims, pooling=pooling, opt=opt, strategy=strategy, learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, dims, pooling=pooling, opt=opt, strategy=strategy, learning_rate=0.2, monkeypatch=monkeypatch)
    rng = np.random.default_rng(5)
    F, B = 3, 700     # ~5 K keys: several 1024-key tiles, many keys shared between tiles
    for it in range(5):
        keys, off = _batch(rng, F, B, 900 + 300 * it)
        ref.train(); dut.train()
        o_ref = ref(keys, off)
        o_dut = dut(keys, off)
        # (the rows differ by the fp32 rounding of a different gradient summation order from the second step on)
        torch.testing.assert_close(o_ref, o_dut, rtol=1e-5, atol=1e-5, msg=f"iteration {it}: forward differs")
        # positive gradients: sums over the occurrences of a row cannot cancel, so the comparison stays well conditioned
        # under a different summation order (Adam's first steps are sign-like: a sum near zero would flip the update)
        g = torch.rand_like(o_ref) + 0.1
        o_ref.backward(g)
        o_dut.backward(g)
        assert torch.equal(ref.size(), dut.size())
    # same keys stored, same rows (fp32 sums of fp32 gradients in a different order: 1e-6 relative)
    for t in range(len(dims)):
        k1, v1 = ref.export_keys_values(ref._table_names[t], torch.device(DEV))
        k2, v2 = dut.export_keys_values(dut._table_names[t], torch.device(DEV))
        o1, o2 = torch.argsort(k1), torch.argsort(k2)
        if not (k1.numel() == k2.numel() and torch.equal(k1[o1], k2[o2])):
            a_, b_ = set(k1.tolist()), set(k2.tolist())
            raise AssertionError(f"table {t}: stored keys differ: {k1.numel()} vs {k2.numel()} exported, sizes "
                                 f"{ref.size(t).item()} vs {dut.size(t).item()}, only per-op {sorted(a_ - b_)[:8]}, "
                                 f"only fused {sorted(b_ - a_)[:8]}, duplicates fused {k2.numel() - len(b_)} per-op {k1.numel() - len(a_)}")
        torch.testing.assert_close(v1[o1], v2[o2], rtol=2e-5, atol=2e-6)
    ref.eval(); dut.eval()
    keys, off = _batch(rng, F, B, 3000)     # known and unknown keys
    with torch.no_grad():
        torch.testing.assert_close(ref(keys, off), dut(keys, off), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("bags,opt,strategy", [(15_000, "SGD", "TIMESTAMP"), (86_000, "ADAM", "LFU"), (87_500, "SGD", "STEP"),
                                               (88_500, "EXACT_ROWWISE_ADAGRAD", "TIMESTAMP"), (225_000, "SGD", "LFU")])
def test_csr_writing_partition_kernel_against_the_per_op_chain_around_its_size_limits(bags, opt, strategy, monkeypatch):
    """path (c) of the fused forward (fused_part3_kernel writes the backward's CSR; reverse indices are materialised on
    demand) against the per-op chain over training steps, at key counts around its switches: just above 64 K keys, either
    side of the 256-partition rule (393 216 keys: above it the partition blocks run in more than one generation), and near
    the 1 M-key end.  Same outputs, same stored keys, same rows; the lazily produced reverse indices equal the eager ones."""
    ref = _mk(False, (16,), cap=1 << 21, opt=opt, strategy=strategy, learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, (16,), cap=1 << 21, opt=opt, strategy=strategy, learning_rate=0.2, monkeypatch=monkeypatch)
    rng = np.random.default_rng(bags)
    ref.train(); dut.train()
    took_c = 0
    for it in range(3):
        lens = rng.integers(1, 9, size=bags)
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(DEV)
        nk = int(off[-1])
        keys = torch.from_numpy(((rng.zipf(1.15, nk) + 7 * it) % (400_000 + 100_000 * it)).astype(np.int64)).to(DEV)
        o_ref, s_ref = ref._forward_impl(keys, off, train=True)
        o_dut, s_dut = dut._forward_impl(keys, off, train=True)
        took_c += int(bool(getattr(s_dut, "lazy", False)))
        torch.testing.assert_close(o_ref, o_dut, rtol=1e-5, atol=1e-5, msg=f"step {it} ({nk} keys): forward differs")
        # the unique ORDER differs between the paths; what must agree is the grouping: unique[reverse] == keys on both
        nu_r, nu_d = int(s_ref.uoff[-1]), int(s_dut.uoff[-1])
        assert nu_r == nu_d
        if it == 1:
            uk = torch.empty(nu_d, dtype=torch.int64, device=DEV)
            rev = s_dut.rev                      # materialises (one kernel) when the step was lazy
            assert int(rev.min()) >= 0 and int(rev.max()) < nu_d
            uk[rev] = keys
            assert torch.equal(uk[rev], keys) and int(torch.unique(rev).numel()) == nu_d
        g = torch.rand_like(o_ref) + 0.1
        ref._backward_impl(s_ref, g)
        dut._backward_impl(s_dut, g)
        assert torch.equal(ref.size(), dut.size())
        assert _counters_clear(dut)
    assert took_c == 3, "the batch did not take the CSR-writing partition path"
    k1, v1 = ref.export_keys_values(ref._table_names[0], torch.device(DEV))
    k2, v2 = dut.export_keys_values(dut._table_names[0], torch.device(DEV))
    o1, o2 = torch.argsort(k1), torch.argsort(k2)
    assert k1.numel() == k2.numel() and torch.equal(k1[o1], k2[o2])
    torch.testing.assert_close(v1[o1], v2[o2], rtol=3e-5, atol=3e-6)
    ref.eval(); dut.eval()
    with torch.no_grad():
        torch.testing.assert_close(ref._forward_impl(keys, off, train=False)[0], dut._forward_impl(keys, off, train=False)[0],
                                   rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("bags,pooling,opt,strategy", [(450_000, "SUM", "SGD", "TIMESTAMP"), (1_350_000, "SUM", "ADAM", "LFU"),
                                                       (2_300_000, "NONE", "SGD", "STEP")])
def test_batches_beyond_a_million_keys_against_the_per_op_chain(bags, pooling, opt, strategy, monkeypatch):
    """Batches beyond 1 M keys (the per-slot-counter index path: the partitioned stage serves 64 K .. 1 M keys; round 5's opt-in
    stage for larger batches measured at parity with this path and was removed in round 6) against the per-op chain at ~2 M and
    ~6 M keys (pooled) and 2.3 M tokens (sequence): same outputs, same unique counts, a consistent reverse index, same stored keys
    and rows after training steps that insert, and steps in the steady state."""
    cap = 1 << 23
    ref = _mk(False, (16,), cap=cap, pooling=pooling, opt=opt, strategy=strategy, learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, (16,), cap=cap, pooling=pooling, opt=opt, strategy=strategy, learning_rate=0.2, monkeypatch=monkeypatch)
    rng = np.random.default_rng(bags)
    ref.train(); dut.train()
    for it in range(3):
        if pooling == "NONE":
            nk = bags
            off = torch.arange(nk + 1, dtype=torch.int64, device=DEV)
        else:
            lens = rng.integers(0, 9, size=bags)
            off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(DEV)
            nk = int(off[-1])
        keys = torch.from_numpy(((rng.zipf(1.1, nk) + 13 * it) % (2_500_000 + 500_000 * min(it, 1))).astype(np.int64)).to(DEV)
        assert nk > (1 << 20)
        o_ref, s_ref = ref._forward_impl(keys, off, train=True)
        o_dut, s_dut = dut._forward_impl(keys, off, train=True)
        torch.testing.assert_close(o_ref, o_dut, rtol=1e-5, atol=1e-5, msg=f"step {it} ({nk} keys): forward differs")
        nu_r, nu_d = int(s_ref.uoff[-1]), int(s_dut.uoff[-1])
        assert nu_r == nu_d
        if it == 1:
            rev = s_dut.rev
            assert int(rev.min()) >= 0 and int(rev.max()) < nu_d
            uk = torch.empty(nu_d, dtype=torch.int64, device=DEV)
            uk[rev] = keys
            assert torch.equal(uk[rev], keys) and int(torch.unique(rev).numel()) == nu_d
        g = torch.rand_like(o_ref) + 0.1
        ref._backward_impl(s_ref, g)
        dut._backward_impl(s_dut, g)
        assert torch.equal(ref.size(), dut.size())
        assert _counters_clear(dut)
    k1, v1 = ref.export_keys_values(ref._table_names[0], torch.device(DEV))
    k2, v2 = dut.export_keys_values(dut._table_names[0], torch.device(DEV))
    o1, o2 = torch.argsort(k1), torch.argsort(k2)
    assert k1.numel() == k2.numel() and torch.equal(k1[o1], k2[o2])
    torch.testing.assert_close(v1[o1], v2[o2], rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("variant", ["1", "2", "3"])
@pytest.mark.parametrize("pooling,bags,bucket", [("SUM", 60_000, 128), ("NONE", 150_000, 128), ("SUM", 40_000, 16), ("SUM", 60_000, 48)])
def test_every_tile_shape_of_the_round5_probe_kernel_against_the_per_op_chain(variant, pooling, bags, bucket, monkeypatch):
    """probe_c_kernel (csrc/probe_c.h; MI355_PROBE_C, a test hook, picks the tile shape) against the per-op chain: pooled and
    sequence lookups, steps that insert every key, steps in the steady state, a 16-slot-bucket table that evicts (deferred keys
    resolved by the partition kernel); 48-slot buckets (not a power of two) take the round-3 probe kernel with the generic bucket
    arithmetic.  Same outputs, same unique counts, same stored keys and rows."""
    monkeypatch.setenv("MI355_PROBE_C", variant)
    cap = 1 << 20 if bucket == 128 else (1 << 16 if bucket == 16 else 48 << 14)
    ref = _mk(False, (16,), cap=cap, pooling=pooling, opt="SGD", strategy="STEP" if bucket == 16 else "TIMESTAMP", bucket=bucket,
              learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, (16,), cap=cap, pooling=pooling, opt="SGD", strategy="STEP" if bucket == 16 else "TIMESTAMP", bucket=bucket,
              learning_rate=0.2, monkeypatch=monkeypatch)
    rng = np.random.default_rng(bags + int(variant))
    ref.train(); dut.train()
    for it in range(3):
        if pooling == "NONE":
            nk = bags
            off = torch.arange(nk + 1, dtype=torch.int64, device=DEV)
        else:
            lens = rng.integers(0, 9, size=bags)          # (empty bags included)
            off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(DEV)
            nk = int(off[-1])
        hi = (200_000 + 50_000 * it) if bucket == 128 else (30_000 + 20_000 * it)
        keys = torch.from_numpy(((rng.zipf(1.2, nk) + 11 * it) % hi).astype(np.int64)).to(DEV)
        o_ref, s_ref = ref._forward_impl(keys, off, train=True)
        o_dut, s_dut = dut._forward_impl(keys, off, train=True)
        assert getattr(s_dut, "lazy", False), "the batch did not take the CSR-writing partition path"
        if bucket == 128:
            torch.testing.assert_close(o_ref, o_dut, rtol=1e-5, atol=1e-5, msg=f"step {it} ({nk} keys): forward differs")
        nu_r, nu_d = int(s_ref.uoff[-1]), int(s_dut.uoff[-1])
        # (keys that find no slot at all in a full 16-slot bucket are one unique row each on the per-op chain and ONE row-less entry
        #  per partition here: no row is updated from either)
        assert nu_r == nu_d if bucket == 128 else nu_d <= nu_r
        rev = s_dut.rev
        assert int(rev.min()) >= 0 and int(rev.max()) < nu_d and int(torch.unique(rev).numel()) == nu_d
        if bucket == 128:
            uk = torch.empty(nu_d, dtype=torch.int64, device=DEV)
            uk[rev] = keys
            assert torch.equal(uk[rev], keys)
        g = torch.rand_like(o_ref) + 0.1
        ref._backward_impl(s_ref, g)
        dut._backward_impl(s_dut, g)
        assert torch.equal(ref.size(), dut.size())
        assert _counters_clear(dut)
    if bucket == 128:      # (an evicting table keeps whichever keys its eviction order chose: sizes agree, contents need not)
        k1, v1 = ref.export_keys_values(ref._table_names[0], torch.device(DEV))
        k2, v2 = dut.export_keys_values(dut._table_names[0], torch.device(DEV))
        o1, o2 = torch.argsort(k1), torch.argsort(k2)
        assert k1.numel() == k2.numel() and torch.equal(k1[o1], k2[o2])
        torch.testing.assert_close(v1[o1], v2[o2], rtol=3e-5, atol=3e-6)


@pytest.mark.parametrize("pooling,opt,strategy", [("NONE", "SGD", "TIMESTAMP"), ("SUM", "ADAM", "LFU"), ("MEAN", "EXACT_ROWWISE_ADAGRAD", "STEP")])
def test_multi_table_batches_of_a_few_hundred_thousand_keys_against_the_per_op_chain(pooling, opt, strategy, monkeypatch):
    """what the HSTU example's embedding collection sends: several tables, sequence or pooled lookups, 10^5 keys per step --
    the per-slot-counter path of the fused forward on grids larger than the resident group, against the per-op chain"""
    dims = (16, 16, 16) if pooling == "NONE" else (8, 16, 32)
    fmap = [0, 1, 1, 2]
    ref = _mk(False, dims, cap=1 << 19, pooling=pooling, opt=opt, strategy=strategy, fmap=fmap, learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, dims, cap=1 << 19, pooling=pooling, opt=opt, strategy=strategy, fmap=fmap, learning_rate=0.2, monkeypatch=monkeypatch)
    rng = np.random.default_rng(11)
    F, B = 4, 12_000
    ref.train(); dut.train()
    for it in range(3):
        lens = rng.integers(0, 9, size=F * B)
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(DEV)
        nk = int(off[-1])
        keys = torch.from_numpy(((rng.zipf(1.2, nk) + 3 * it) % (150_000 + 40_000 * it)).astype(np.int64)).to(DEV)
        o_ref, o_dut = ref(keys, off), dut(keys, off)
        torch.testing.assert_close(o_ref, o_dut, rtol=1e-5, atol=1e-5, msg=f"step {it} ({nk} keys): forward differs")
        g = torch.rand_like(o_ref) + 0.1
        o_ref.backward(g); o_dut.backward(g)
        assert torch.equal(ref.size(), dut.size())
        assert _counters_clear(dut)
    for t in range(len(dims)):
        k1, v1 = ref.export_keys_values(ref._table_names[t], torch.device(DEV))
        k2, v2 = dut.export_keys_values(dut._table_names[t], torch.device(DEV))
        o1, o2 = torch.argsort(k1), torch.argsort(k2)
        assert k1.numel() == k2.numel() and torch.equal(k1[o1], k2[o2])
        torch.testing.assert_close(v1[o1], v2[o2], rtol=3e-5, atol=3e-6)


@pytest.mark.parametrize("case", ["eight_equal", "skewed", "shared_table_mixed_dims", "seventy_tables"])
@pytest.mark.parametrize("pooling,opt,strategy", [("SUM", "SGD", "TIMESTAMP"), ("MEAN", "ADAM", "LFU"), ("SUM", "EXACT_ROWWISE_ADAGRAD", "STEP"),
                                                  ("NONE", "SGD", "LFU"), ("NONE", "ADAM", "TIMESTAMP")])
def test_multi_table_pooled_batches_take_the_csr_writing_partition_path(case, pooling, opt, strategy, monkeypatch):
    """round 4: path (c) with TABLE-ALIGNED partitions (every table owns whole partitions, the rest dealt by its share of the
    batch's keys) against the per-op chain -- eight equal tables (the 8 x 8192-bag shape of tools/bench_model_shapes.py), a skewed
    batch (one table with nearly all keys, one with a handful, one with NONE), and two features sharing a table with mixed row
    widths; pooled (SUM / MEAN) and sequence lookups (NONE: occurrence j is its own bag, rows copied by the late-row variant of the
    sequence gather).  Same output, the same unique rows PER TABLE (the unique order stays table-major), reverse indices that point
    at the key's own table, same stored keys / rows / optimizer state after three steps."""
    if pooling == "NONE" and case == "shared_table_mixed_dims":
        pytest.skip("sequence lookups need one row width")
    if case == "eight_equal":
        dims, fmap, B = (16,) * 8, None, 6_000
        hi = [60_000] * 8
        maxlen = [9] * 8
    elif case == "skewed":
        dims, fmap, B = (16, 16, 16, 16), None, 30_000
        hi = [300_000, 50, 1_000, 7]
        maxlen = [9, 2, 0, 2]           # feature 2 has no key at all; features 1 and 3 a few thousand
    elif case == "seventy_tables":     # more tables than a wave has lanes: the partition counts are scanned two per lane
        dims, fmap, B = (16,) * 70, None, 1_500
        hi = [4_000 + 50 * t for t in range(70)]
        maxlen = [9] * 70
    else:
        dims, fmap, B = (8, 32, 16), [0, 1, 1, 2], 12_000
        hi = [50_000, 80_000, 80_000, 20_000]
        maxlen = [9, 9, 5, 9]
    F = len(hi)
    cap = 1 << 14 if case == "seventy_tables" else 1 << 19
    ref = _mk(False, dims, cap=cap, pooling=pooling, opt=opt, strategy=strategy, fmap=fmap, learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, dims, cap=cap, pooling=pooling, opt=opt, strategy=strategy, fmap=fmap, learning_rate=0.2, monkeypatch=monkeypatch)
    fm = fmap or list(range(F))
    rng = np.random.default_rng(5)
    ref.train(); dut.train()
    took_c = 0
    for it in range(3):
        lens = np.concatenate([rng.integers(0, m, size=B) if m > 0 else np.zeros(B, np.int64) for m in maxlen])
        off_np = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nk = int(off_np[-1])
        keys_np = np.empty(nk, np.int64)
        tab_of = np.empty(nk, np.int64)
        for f in range(F):
            lo, hi_ = off_np[f * B], off_np[(f + 1) * B]
            keys_np[lo:hi_] = (rng.zipf(1.2, hi_ - lo) + 5 * it) % (hi[f] + 1000 * it)
            tab_of[lo:hi_] = fm[f]
        keys, off = torch.from_numpy(keys_np).to(DEV), torch.from_numpy(off_np).to(DEV)
        o_ref, s_ref = ref._forward_impl(keys, off, train=True)
        o_dut, s_dut = dut._forward_impl(keys, off, train=True)
        took_c += int(bool(getattr(s_dut, "lazy", False)))
        torch.testing.assert_close(o_ref, o_dut, rtol=1e-5, atol=1e-5, msg=f"step {it} ({nk} keys): forward differs")
        assert torch.equal(s_ref.uoff.cpu(), s_dut.uoff.cpu()), "unique rows per table differ"
        nu = int(s_dut.uoff[-1])
        if it == 1:
            rev = s_dut.rev
            assert int(rev.min()) >= 0 and int(rev.max()) < nu
            tids = s_dut.tids[:nu]
            assert torch.equal(tids[rev].cpu(), torch.from_numpy(tab_of)), "a reverse index points into another table's unique rows"
            # table-major unique order: the table ids are non-decreasing and agree with the table offsets
            uo = s_dut.uoff.cpu().numpy()
            t_np = tids.cpu().numpy()
            for t in range(len(dims)):
                assert (t_np[uo[t]:uo[t + 1]] == t).all()
            uk = torch.zeros(nu, dtype=torch.int64, device=DEV)
            uk[rev] = keys
            assert torch.equal(uk[rev], keys)
            assert int(torch.unique(rev).numel()) == nu
        g = torch.rand_like(o_ref) + 0.1
        ref._backward_impl(s_ref, g)
        dut._backward_impl(s_dut, g)
        assert torch.equal(ref.size(), dut.size())
        assert _counters_clear(dut)
    assert took_c == 3, "the multi-table batch did not take the CSR-writing partition path"
    for t in range(len(dims)):
        k1, v1 = ref.export_keys_values(ref._table_names[t], torch.device(DEV))
        k2, v2 = dut.export_keys_values(dut._table_names[t], torch.device(DEV))
        o1, o2 = torch.argsort(k1), torch.argsort(k2)
        assert k1.numel() == k2.numel() and torch.equal(k1[o1], k2[o2])
        torch.testing.assert_close(v1[o1], v2[o2], rtol=3e-5, atol=3e-6)
    # eval forward (round 4: ONE kernel for several tables and for sequence lookups too -- every lane probes its own key, the
    # key's table follows from its position) against the per-op chain; unknown keys (the shifted stream) give zero rows
    ref.eval(); dut.eval()
    with torch.no_grad():
        for kk in (keys, keys + 1_000_003):
            torch.testing.assert_close(ref._forward_impl(kk, off, train=False)[0], dut._forward_impl(kk, off, train=False)[0],
                                       rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tokens,opt,strategy,bucket", [(70_000, "SGD", "TIMESTAMP", 128), (131_072, "ADAM", "LFU", 128), (400_000, "SGD", "STEP", 128),
                                                         (90_000, "SGD", "LFU", 16)])
def test_sequence_lookups_of_one_table_take_the_csr_writing_partition_path(tokens, opt, strategy, bucket, monkeypatch):
    """round 4: sequence lookups (pooling NONE) on path (c) -- the probe kernel takes occurrence j as its own bag, the partition
    kernel writes the backward's CSR over gradient ROWS, the rows are copied by gather_rows_late_kernel -- against the per-op chain
    over training steps (the last case with 16-slot buckets that fill up: rows that come out of an eviction are resolved late,
    through the key's record; there only what does not depend on the eviction order is compared)."""
    cap = 1 << 21 if bucket == 128 else 1 << 15
    ref = _mk(False, (16,), cap=cap, pooling="NONE", opt=opt, strategy=strategy, bucket=bucket, learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, (16,), cap=cap, pooling="NONE", opt=opt, strategy=strategy, bucket=bucket, learning_rate=0.2, monkeypatch=monkeypatch)
    rng = np.random.default_rng(tokens)
    ref.train(); dut.train()
    took_c = 0
    exact = bucket == 128
    for it in range(3):
        off = torch.arange(tokens + 1, dtype=torch.int64, device=DEV)
        hi = 300_000 + 100_000 * it if exact else 60_000
        keys = torch.from_numpy(((rng.zipf(1.15, tokens) + 7 * it) % hi).astype(np.int64)).to(DEV)
        o_ref, s_ref = ref._forward_impl(keys, off, train=True)
        o_dut, s_dut = dut._forward_impl(keys, off, train=True)
        took_c += int(bool(getattr(s_dut, "lazy", False)))
        # (keys that find no slot at all -- every slot of their bucket is used by this very batch -- are one unique row each on the
        #  per-op chain and ONE row-less entry per partition here: no row is updated from either)
        assert int(s_ref.uoff[-1]) == int(s_dut.uoff[-1]) if exact else int(s_dut.uoff[-1]) <= int(s_ref.uoff[-1])
        if exact:
            torch.testing.assert_close(o_ref, o_dut, rtol=1e-5, atol=1e-5, msg=f"step {it}: forward differs")
        else:   # every occurrence of a key got the same row, and it is the row the table holds for the key right now
            ks, vs = dut.export_keys_values(dut._table_names[0], torch.device(DEV))
            order = torch.argsort(ks)
            ks, vs = ks[order], vs[order]
            idx = torch.searchsorted(ks, keys).clamp(max=ks.numel() - 1)
            found = ks[idx] == keys
            torch.testing.assert_close(o_dut, vs[idx] * found[:, None], rtol=1e-6, atol=1e-6)
            assert int(found.sum()) > tokens // 2
        if it == 1:
            nu = int(s_dut.uoff[-1])
            rev = s_dut.rev
            uk = torch.zeros(nu, dtype=torch.int64, device=DEV)
            uk[rev] = keys
            has_row = torch.ones_like(keys, dtype=torch.bool) if exact else found   # (row-less keys share an entry, see above)
            assert torch.equal(uk[rev][has_row], keys[has_row]) and int(torch.unique(rev).numel()) == nu
        g = torch.rand_like(o_ref) + 0.1
        ref._backward_impl(s_ref, g)
        dut._backward_impl(s_dut, g)
        assert torch.equal(ref.size(), dut.size())
        assert _counters_clear(dut)
    assert took_c == 3, "the sequence batch did not take the CSR-writing partition path"
    ref.eval(); dut.eval()
    with torch.no_grad():   # one-kernel sequence eval (gather_rows_eval_kernel): known keys give their rows, unknown ones zeros
        for kk in (keys, keys + 1_000_003):
            e_dut = dut._forward_impl(kk, off, train=False)[0]
            if exact:
                torch.testing.assert_close(ref._forward_impl(kk, off, train=False)[0], e_dut, rtol=1e-5, atol=1e-5)
            else:
                ks, vs = dut.export_keys_values(dut._table_names[0], torch.device(DEV))
                order = torch.argsort(ks)
                ks, vs = ks[order], vs[order]
                idx = torch.searchsorted(ks, kk).clamp(max=ks.numel() - 1)
                torch.testing.assert_close(e_dut, vs[idx] * (ks[idx] == kk)[:, None], rtol=1e-6, atol=1e-6)
    if exact:
        k1, v1 = ref.export_keys_values(ref._table_names[0], torch.device(DEV))
        k2, v2 = dut.export_keys_values(dut._table_names[0], torch.device(DEV))
        o1, o2 = torch.argsort(k1), torch.argsort(k2)
        assert k1.numel() == k2.numel() and torch.equal(k1[o1], k2[o2])
        torch.testing.assert_close(v1[o1], v2[o2], rtol=3e-5, atol=3e-6)


def test_multi_table_partition_path_with_full_buckets(monkeypatch):
    """several small tables whose buckets fill up: the keys without a free slot go through the eviction inside the partition
    block of THEIR table (row address, table-relative slot and row initialisation all come from the partition's table).  Which
    of two equal scores is evicted is schedule dependent, so the per-op chain is only held to what is not (unique rows per table,
    table sizes); the pooled output of every training forward must equal the pooled rows of the module's OWN tables as exported
    right after it (a late row resolved against the wrong table, or initialised in it, shows up there)."""
    dims = (16, 16, 16, 16)
    ref = _mk(False, dims, cap=16_384, pooling="SUM", opt="SGD", strategy="LFU", bucket=16, learning_rate=0.2, monkeypatch=monkeypatch)
    dut = _mk(True, dims, cap=16_384, pooling="SUM", opt="SGD", strategy="LFU", bucket=16, learning_rate=0.2, monkeypatch=monkeypatch)
    rng = np.random.default_rng(9)
    F, B = 4, 8_000
    ref.train(); dut.train()
    took_c = 0
    for it in range(4):
        lens = rng.integers(1, 6, size=F * B)
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(DEV)
        nk = int(off[-1])
        keys = torch.from_numpy(((rng.zipf(1.3, nk) + 11 * it) % 40_000).astype(np.int64)).to(DEV)
        o_ref, s_ref = ref._forward_impl(keys, off, train=True)
        o_dut, s_dut = dut._forward_impl(keys, off, train=True)
        took_c += int(bool(getattr(s_dut, "lazy", False)))
        assert torch.equal(s_ref.uoff.cpu(), s_dut.uoff.cpu())
        bag_of = torch.repeat_interleave(torch.arange(F * B, device=DEV), off[1:] - off[:-1])
        for f in range(F):
            ks, vs = dut.export_keys_values(dut._table_names[f], torch.device(DEV))
            order = torch.argsort(ks)
            ks, vs = ks[order], vs[order][:, :16].float()
            lo, hi = int(off[f * B]), int(off[(f + 1) * B])
            kf = keys[lo:hi]
            idx = torch.searchsorted(ks, kf).clamp(max=ks.numel() - 1)
            found = ks[idx] == kf
            rows = vs[idx] * found[:, None]
            want = torch.zeros(B, 16, device=DEV).index_add_(0, bag_of[lo:hi] - f * B, rows)
            torch.testing.assert_close(o_dut[:, 16 * f: 16 * (f + 1)], want, rtol=1e-5, atol=1e-5,
                                       msg=f"step {it}, table {f}: the pooled output is not the sum of the table's own rows")
        assert int(found.sum()) > 0
        g = torch.rand_like(o_ref) + 0.1
        ref._backward_impl(s_ref, g)
        dut._backward_impl(s_dut, g)
        assert torch.equal(ref.size(), dut.size())
        assert _counters_clear(dut)
    assert took_c == 4


def test_one_new_key_in_every_tile_gets_one_slot(monkeypatch):
    """cold start: the same unseen keys arrive from dozens of tiles at once; every key must end in exactly one slot and
    the reverse indices must group all its occurrences"""
    m = _mk(True, (8,), cap=1 << 16, pooling="NONE", monkeypatch=monkeypatch)
    rng = np.random.default_rng(0)
    hot = rng.integers(0, 1 << 40, size=50).astype(np.int64)
    keys = np.concatenate([hot[rng.integers(0, 50, size=60_000)], rng.integers(0, 1 << 40, size=20_000)]).astype(np.int64)
    rng.shuffle(keys)
    kt = torch.from_numpy(keys).to(DEV)
    off = torch.arange(keys.size + 1, dtype=torch.int64, device=DEV)
    m.train()
    out, st = m._forward_impl(kt, off, train=True)
    nu = int(st.uoff[-1])
    assert nu == np.unique(keys).size == int(m.size())
    rev = st.rev
    # occurrences of one key share one unique id, distinct keys never do
    first = torch.full((nu,), -1, dtype=torch.int64, device=DEV)
    first[rev] = kt
    assert torch.equal(first[rev], kt)
    assert torch.equal(torch.bincount(rev, minlength=nu).to(torch.int32), st.csr_cnt[:nu])
    # sequence output row j == the stored row of key j
    found, rows = m.lookup_rows(kt, 0)
    assert bool(found.all()) and torch.equal(out, rows[:, :8])
    # the scratch counters are back to zero
    assert _counters_clear(m)
    m._backward_impl(st, torch.zeros_like(out))


@pytest.mark.parametrize("strategy", ["LFU", "CUSTOMIZED", "STEP"])
@pytest.mark.parametrize("fused", [True, False])
def test_hits_and_misses_in_a_full_bucket(strategy, fused, monkeypatch):
    """one bucket, full; a batch of keys it holds plus new keys: the new keys may only evict keys that are NOT in the batch,
    every key of the batch reads its own row, and the backward updates every row once"""
    m = _mk(fused, (8,), cap=128, pooling="NONE", strategy=strategy, learning_rate=1.0, monkeypatch=monkeypatch)
    m.train()
    if strategy == "CUSTOMIZED":
        m.set_score(7)      # constant score: ties everywhere
    old = torch.arange(1000, 1128, dtype=torch.int64, device=DEV)
    off = lambda n: torch.arange(n + 1, dtype=torch.int64, device=DEV)
    m(old, off(128))
    assert int(m.size()) == 128
    _, rows_old = m.lookup_rows(old, 0)
    hits = old[:96]                                   # the LOWEST slots are the tie-break's first victims
    new = torch.arange(5000, 5032, dtype=torch.int64, device=DEV)
    batch = torch.cat([hits, new])
    out, st = m._forward_impl(batch, off(128), train=True)
    # every hit key reads the row it had before; nothing of the batch was evicted
    assert torch.equal(out[:96], rows_old[:96, :8])
    found, rows_now = m.lookup_rows(batch, 0)
    assert bool(found.all())
    assert torch.equal(out, rows_now[:, :8])
    # exactly the 32 keys outside the batch made room
    found_rest, _ = m.lookup_rows(old[96:], 0)
    assert int(found_rest.sum()) == 0 and int(m.size()) == 128
    # one SGD step with a gradient of ones: every row of the batch moves by exactly -1
    m._backward_impl(st, torch.ones_like(out))
    _, rows_after = m.lookup_rows(batch, 0)
    torch.testing.assert_close(rows_after[:, :8], rows_now[:, :8] - 1.0, rtol=0, atol=1e-6)


def _partitions(m, n):
    from mi355_native import lib
    return int(lib().mi355_demb_forward_fused_partitions(n, m.num_tables, m.table.num_buckets_))


@pytest.mark.parametrize("strategy", ["STEP", "LFU"])
@pytest.mark.parametrize("bucket,n", [(128, 400_000), (16, 100_000)], ids=["per_slot_counters", "slot_range_partitions"])
def test_many_deferred_keys_in_a_grid_larger_than_the_resident_group(bucket, n, strategy, monkeypatch):
    """a full table, a 400 K-key batch of old and new keys: thousands of keys find their bucket full and are deferred to the
    head of the numbering kernel, whose grid (391 blocks) is larger than the group of blocks that runs the eviction (one
    per CU) -- the later blocks wait for the release flag.  Checks: every key of the batch that has a slot reads its own
    row; keys of the batch are never evicted by the batch; the size stays at capacity; unique[reverse] == keys; the
    scratch counters are clean; the backward moves every row of the batch exactly once."""
    cap = 64 * 1024
    m = _mk(True, (8,), cap=cap, pooling="NONE", strategy=strategy, learning_rate=1.0, bucket=bucket, monkeypatch=monkeypatch)
    m.train()
    # the second configuration (4096 small buckets) takes the partitioned index stage: the deferred keys are evicted for
    # inside the block that owns their slot range (fused_part_kernel)
    assert (_partitions(m, n) > 0) == (bucket == 16)
    off = lambda n: torch.arange(n + 1, dtype=torch.int64, device=DEV)
    rng = np.random.default_rng(11)
    old = torch.from_numpy(rng.permutation(1 << 22)[: 2 * cap].astype(np.int64)).to(DEV)
    for i in range(0, old.numel(), 32768):                 # fill: the table ends up full (evicting among `old` itself)
        m(old[i:i + 32768], off(min(32768, old.numel() - i)))
    full = int(m.size())
    assert cap - 64 <= full <= cap           # (with 16-slot buckets a handful of buckets see fewer than 16 of the fill keys)
    f_old, _ = m.lookup_rows(old, 0)
    resident = old[f_old]
    new = torch.arange(1 << 23, (1 << 23) + 20_000, dtype=torch.int64, device=DEV)       # 20 K keys the table has never seen
    pool = torch.cat([resident[: 30_000], new])
    batch = pool[torch.from_numpy(rng.integers(0, pool.numel(), n)).to(DEV)]
    _, rows_before = m.lookup_rows(resident[: 30_000], 0)
    out, st = m._forward_impl(batch, off(n), train=True)
    nu = int(st.uoff[-1])
    assert torch.equal(st.unique_keys[:nu][st.rev], batch) if hasattr(st, "unique_keys") else True
    found, rows_now = m.lookup_rows(batch, 0)
    has = st.slots[:nu][st.rev] >= 0
    assert bool(found[has].all()) and torch.equal(out[has], rows_now[has][:, :8])
    assert bool((out[~has] == 0).all())
    assert int(has.sum()) > n * 0.8 and full <= int(m.size()) <= cap   # (a bucket whose every slot the batch uses refuses the rest)
    # the 30 K resident keys of the pool were hits: same rows as before, none evicted
    f_res, rows_res = m.lookup_rows(resident[: 30_000], 0)
    in_batch = torch.isin(resident[: 30_000], batch)
    assert bool(f_res[in_batch].all()) and torch.equal(rows_res[in_batch], rows_before[in_batch])
    # new keys did come in
    f_new, _ = m.lookup_rows(new, 0)
    assert int(f_new.sum()) > (15_000 if bucket == 128 else 8_000)   # (16-slot buckets fill up with the batch's own keys sooner)
    m._backward_impl(st, torch.ones_like(out))
    assert _counters_clear(m)
    _, rows_after = m.lookup_rows(batch, 0)
    cnt = torch.zeros(nu, device=DEV).index_add_(0, st.rev, torch.ones(n, device=DEV))[st.rev]
    torch.testing.assert_close(rows_after[has][:, :8], rows_now[has][:, :8] - cnt[has][:, None], rtol=0, atol=1e-4)


def _fmix64(k):
    k = k.astype(np.uint64)
    k ^= k >> np.uint64(33); k *= np.uint64(0xFF51AFD7ED558CCD)
    k ^= k >> np.uint64(33); k *= np.uint64(0xC4CEB9FE1A85EC53)
    k ^= k >> np.uint64(33)
    return k


@pytest.mark.parametrize("strategy,opt", [("TIMESTAMP", "SGD"), ("LFU", "ADAM")])
def test_partitioned_stage_matches_the_per_slot_counter_path_and_flags_a_flooded_partition(strategy, opt, monkeypatch):
    """(i) the partitioned index stage (one table, >= 64 K keys) and the per-slot-counter path (MI355_FUSED_PART=0) give the
    same pooled output, the same rows after a training step and the same table; (ii) a key stream built to land in ONE slot
    range floods that partition's record list: the surplus keys take no part in that step's bookkeeping, and the module
    reports the sticky flag instead of training on silently."""
    cap, C, n = 1 << 20, 128, 80_000
    a = _mk(True, (16,), cap=cap, pooling="SUM", learning_rate=0.5, strategy=strategy, opt=opt, monkeypatch=monkeypatch)
    assert _partitions(a, n) > 0
    rng = np.random.default_rng(2)
    lens = rng.integers(1, 9, size=n // 4)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(DEV)
    nk = int(off[-1])
    keys = torch.from_numpy((rng.zipf(1.2, nk) % 200_000).astype(np.int64)).to(DEV)
    assert _partitions(a, nk) > 0
    a.train()
    out_a = a(keys, off)
    out_a.backward(torch.ones_like(out_a))
    import subprocess, sys, os, json
    # the other path needs its own process (the switch is read once per process): same seed, same batch
    code = f"""
import sys, json, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r}); sys.path.insert(0, {os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'recsys-examples_amd')!r})
import pytest
from test_fused_fwd_gpu import _mk
class MP:
    def setenv(self, k, v):
        import os; os.environ[k] = v
m = _mk(True, (16,), cap={cap}, pooling="SUM", learning_rate=0.5, strategy={strategy!r}, opt={opt!r}, monkeypatch=MP())
keys = torch.from_numpy(np.load(sys.argv[1])).cuda(); off = torch.from_numpy(np.load(sys.argv[2])).cuda()
m.train(); out = m(keys, off); out.backward(torch.ones_like(out))
uk = torch.unique(keys); f, rows = m.lookup_rows(uk, 0)
np.save(sys.argv[3], out.detach().cpu().numpy()); np.save(sys.argv[4], rows.cpu().numpy()); print(int(m.size()), bool(f.all()))
"""
    import tempfile
    d = tempfile.mkdtemp()
    np.save(d + "/k.npy", keys.cpu().numpy()); np.save(d + "/o.npy", off.cpu().numpy())
    env = dict(os.environ, MI355_FUSED_PART="0", MI355_FUSED="1")
    r = subprocess.run([sys.executable, "-c", code, d + "/k.npy", d + "/o.npy", d + "/out.npy", d + "/rows.npy"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    size_b, all_found = r.stdout.split()[-2:]
    uk = torch.unique(keys)
    f, rows = a.lookup_rows(uk, 0)
    assert bool(f.all()) and all_found == "True" and int(a.size()) == int(size_b)
    torch.testing.assert_close(out_a.detach().cpu(), torch.from_numpy(np.load(d + "/out.npy")), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rows.cpu(), torch.from_numpy(np.load(d + "/rows.npy")), rtol=1e-5, atol=1e-5)
    assert _counters_clear(a)
    # (ii) flood one slot range: keys whose bucket lies in partition 0
    P = _partitions(a, n)
    S = a.table.capacity_
    spp = -(-((S + 1 + P - 1) // P) // C) * C
    cand = np.arange(1 << 30, (1 << 30) + 120 * n, dtype=np.int64)
    h = _fmix64(cand) & np.uint64(0x7FFFFFFFFFFFFFFF)
    bucket = (h % np.uint64(S)) // np.uint64(C)
    flood = cand[(bucket * np.uint64(C)) // np.uint64(spp) == 0][:n]
    assert flood.size == n
    fk = torch.from_numpy(flood).to(DEV)
    foff = torch.arange(n + 1, dtype=torch.int64, device=DEV)
    out = a(fk, foff)
    torch.cuda.synchronize()
    assert int(a._fused_aux[5]) == 0 and int(a._fused_aux[6]) != 0      # the step's epoch, not a sticky flag
    # round 6: the step's CSR is incomplete -- its backward first regroups it on the per-slot-counter path (the host reads the
    # step's overflow notice when the backward is issued), then updates every row the step found: nothing is skipped
    probe_keys = torch.cat([fk[:4096], keys[:4096]])
    f_before, rows_before = a.lookup_rows(probe_keys, 0)
    reruns = getattr(a, "overflow_reruns", 0)
    out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    assert getattr(a, "overflow_reruns", 0) == reruns + 1
    f_after, rows_after = a.lookup_rows(probe_keys, 0)
    both = (f_before & f_after)[:4096]
    assert int(both.sum()) > 300
    moved = (rows_before[:4096][both] != rows_after[:4096][both]).any(dim=1)
    assert float(moved.float().mean()) > 0.95, "rows of the flooded step's keys were not updated"
    if opt == "SGD":      # every key of the flood occurs once, gradient 1, lr 0.5
        d = (rows_before[:4096][both] - rows_after[:4096][both])[moved][:, :16]
        torch.testing.assert_close(d, torch.full_like(d, 0.5), rtol=0, atol=1e-5)
    keep = (f_before & f_after)[4096:]                                # keys that were not in the step: untouched
    assert int(keep.sum()) > 2000 and torch.equal(rows_before[4096:][keep], rows_after[4096:][keep])
    served = (out.abs().sum(1) > 0)
    assert 2048 <= int(served.sum()) < n          # (the slot range holds ~13 K rows)
    assert _counters_clear_except_flag(a)
    # and the module keeps working
    out2 = a(keys, off)
    out2.backward(torch.ones_like(out2))
    torch.cuda.synchronize()
    assert int(a._fused_aux[5]) == 0 and bool(torch.isfinite(out2).all())
    assert getattr(a, "overflow_reruns", 0) == reruns + 1


@pytest.mark.parametrize("mode,pooling", [("notice", "SUM"), ("notice", "NONE"), ("notice-untouched", "SUM"), ("1", "SUM"), ("1", "NONE")])
def test_overflowed_partition_is_rerun_and_no_update_is_lost(monkeypatch, tmp_path, mode, pooling):
    """(sequence lookups -- pooling NONE -- since round 5: the partition blocks ride in the gather's launch there.)
    The reference never skips an update (unique_op.cu:484-714).  Round 6, the DEFAULT ("notice"): the partition kernel tells the
    host through pinned memory that a list flooded; the step's backward -- or whoever reads the step's numbering first -- regroups
    it on the per-slot-counter path ("notice-untouched": nobody looks at the step between forward and backward, the plan's
    backward call finds out itself).  MI355_FUSED_OVERFLOW_RERUN=1: the round-5 form, three gated launches behind the gather
    (what a forward captured into a graph uses).  The flood: a batch whose (tile, key)
    records flood ONE slot range -- 4 000 distinct keys of partition 0 drawn 80 000 times: ~64 K records for a list of 2 048, but
    no bucket overfull, so nothing depends on eviction order -- is re-run on the per-slot-counter path inside the same C call:
    the three gated launches behind the gather find the epoch in aux[6] and redo the numbering and the CSR.  Compared with a process
    that runs the per-slot-counter path throughout (MI355_FUSED_PART=0): same outputs of the flooded step and of the steps around
    it, same rows after their backwards, same table size; no sticky flag, nothing raised."""
    import os
    import subprocess
    import sys

    cap, C, n = 1 << 20, 128, 80_000
    a = _mk(True, (16,), cap=cap, pooling=pooling, learning_rate=0.5, strategy="TIMESTAMP", opt="SGD", monkeypatch=monkeypatch)
    P = _partitions(a, n)
    assert P > 0
    S = a.table.capacity_
    spp = -(-((S + 1 + P - 1) // P) // C) * C
    cand = np.arange(1 << 30, (1 << 30) + 6 * 4000 * P, dtype=np.int64)
    h = _fmix64(cand) & np.uint64(0x7FFFFFFFFFFFFFFF)
    bucket = (h % np.uint64(S)) // np.uint64(C)
    pool = cand[(bucket * np.uint64(C)) // np.uint64(spp) == 0][:4000]
    assert pool.size == 4000
    rng = np.random.default_rng(3)
    flood = pool[rng.integers(0, 4000, n)]
    lens = rng.integers(1, 9, size=n // 4)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    keys = (rng.zipf(1.2, int(off[-1])) % 200_000).astype(np.int64)
    d = str(tmp_path)
    np.save(d + "/k.npy", keys); np.save(d + "/o.npy", off); np.save(d + "/f.npy", flood)
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r}); sys.path.insert(0, {os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'recsys-examples_amd')!r})
from test_fused_fwd_gpu import _mk
class MP:
    def setenv(self, k, v):
        import os; os.environ[k] = v
d, tag = sys.argv[1], sys.argv[2]
m = _mk(True, (16,), cap={cap}, pooling={pooling!r}, learning_rate=0.5, strategy="TIMESTAMP", opt="SGD", monkeypatch=MP())
keys = torch.from_numpy(np.load(d + "/k.npy")).cuda(); off = torch.from_numpy(np.load(d + "/o.npy")).cuda()
fk = torch.from_numpy(np.load(d + "/f.npy")).cuda(); foff = torch.arange(fk.numel() + 1, dtype=torch.int64, device="cuda")
m.train()
outs = []
lazy = []
for kk, oo in ((keys, off), (fk, foff), (keys, off)):
    out, st = m._forward_impl(kk, oo, train=True)
    lazy.append(bool(getattr(st, "lazy", False)))
    if tag != "notice-untouched":
        nu = int(st.uoff[-1])
        rev = st.rev
        assert int(rev.min()) >= 0 and int(rev.max()) < nu
    m._backward_impl(st, torch.ones_like(out))
    nu = int(st.uoff[-1])
    outs.append(out.float().cpu().numpy()); outs.append(np.array([nu]))
probe = torch.unique(torch.cat([fk, keys]))
f, rows = m.lookup_rows(probe, 0)
torch.cuda.synchronize()
np.savez(d + "/res_" + tag + ".npz", *outs, found=f.cpu().numpy(), rows=rows.cpu().numpy(), size=int(m.size()), aux5=int(m._fused_aux[5]),
         aux6=int(m._fused_aux[6]), lazy=np.array(lazy), reruns=int(getattr(m, "overflow_reruns", 0)))
"""
    res = {}
    for tag, env in ((mode if mode != "1" else "rerun", dict(MI355_FUSED_OVERFLOW_RERUN="1") if mode == "1" else {}),
                     ("counters", dict(MI355_FUSED_PART="0"))):
        r = subprocess.run([sys.executable, "-c", code, d, tag], env=dict(os.environ, MI355_FUSED="1", **env), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        res[tag] = np.load(d + "/res_" + tag + ".npz")
    A, B = res[mode if mode != "1" else "rerun"], res["counters"]
    assert A["lazy"].tolist() == [True, True, True] and B["lazy"].tolist() == [False, False, False]
    assert int(A["reruns"]) == (0 if mode == "1" else 1) and int(B["reruns"]) == 0
    assert int(A["aux5"]) == 0 and int(A["aux6"]) != 0, "no overflow was seen (or it left the sticky flag)"
    for i in range(6):
        np.testing.assert_allclose(A[f"arr_{i}"], B[f"arr_{i}"], rtol=1e-5, atol=1e-5, err_msg=f"output / unique count {i}")
    assert int(A["size"]) == int(B["size"]) and bool(A["found"].all()) and bool(B["found"].all())
    np.testing.assert_allclose(A["rows"], B["rows"], rtol=1e-4, atol=1e-5)


def _counters_clear_except_flag(m):
    torch.cuda.synchronize()
    aux = m._fused_aux.clone()
    aux[5] = 0
    aux[6] = 0         # (the epoch of the last flooded step: a value, not a flag -- nothing reads it as state)
    cap = m.table.capacity_
    H = 64 + 4096
    return int(aux[:H].abs().sum()) == 0 and int(aux[H: H + 2 * (cap + 1): 2].abs().sum()) == 0


def test_full_bucket_without_a_victim_reports_no_slot(monkeypatch):
    """every slot of the bucket is used by the batch itself: the extra keys get no slot (index -1, zero rows, no update),
    exactly like an insert that returns Busy"""
    m = _mk(True, (8,), cap=128, pooling="NONE", strategy="STEP", monkeypatch=monkeypatch)
    m.train()
    off = lambda n: torch.arange(n + 1, dtype=torch.int64, device=DEV)
    old = torch.arange(1000, 1128, dtype=torch.int64, device=DEV)
    m(old, off(128))
    batch = torch.cat([old, torch.arange(7000, 7010, dtype=torch.int64, device=DEV)])
    out, st = m._forward_impl(batch, off(138), train=True)
    assert bool((out[128:] == 0).all()) and int(m.size()) == 128
    found, rows = m.lookup_rows(old, 0)
    assert bool(found.all()) and torch.equal(out[:128], rows[:, :8])
    nu = int(st.uoff[-1])
    assert int((st.slots[:nu] < 0).sum()) == 1      # the keys without a slot share one placeholder entry
    m._backward_impl(st, torch.ones_like(out))
    assert _counters_clear(m)


def test_fused_forward_from_another_thread_backward(monkeypatch):
    """autograd runs the backward of a CUDA node on its own host thread: the side-stream join must not depend on
    thread-local state (round-1 advisor finding)"""
    import threading

    m = _mk(True, (16,), cap=1 << 14, pooling="SUM", learning_rate=1.0, monkeypatch=monkeypatch)
    m.train()
    rng = np.random.default_rng(2)
    keys, off = _batch(rng, 1, 4000, 3000)
    out, st = m._forward_impl(keys, off, train=True)
    _, rows0 = m.lookup_rows(torch.unique(keys), 0)
    err = []

    def bw():
        try:
            torch.cuda.set_device(0)
            m._backward_impl(st, torch.ones_like(out))
            torch.cuda.synchronize()
        except Exception as e:  # pragma: no cover
            err.append(e)

    t = threading.Thread(target=bw)
    t.start(); t.join()
    assert not err
    uk, cnt = torch.unique(keys, return_counts=True)
    _, rows1 = m.lookup_rows(uk, 0)
    torch.testing.assert_close(rows1[:, :16], rows0[:, :16] - cnt[:, None].float(), rtol=0, atol=1e-4)
