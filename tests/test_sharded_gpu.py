"""GPU tests of the sharded path's HIP pieces (bag permutation, partial-sum reduction) and of the whole
RowWiseShardedLookup with the HIP backend on a 1-rank RCCL group (the W>1 routing is covered on CPU over
gloo in test_sharded_cpu.py; two ranks cannot share the single GPU of the test box)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _perm_ref(S, F, B, lengths, data):
    off = np.zeros(lengths.size + 1, np.int64)
    off[1:] = np.cumsum(lengths)
    out_len = lengths.reshape(S, F, B).transpose(1, 0, 2).reshape(-1)
    chunks = []
    for f in range(F):
        for s in range(S):
            for b in range(B):
                i = s * F * B + f * B + b
                chunks.append(data[off[i]:off[i + 1]])
    out = np.concatenate(chunks) if chunks else data[:0]
    return out_len, out


@pytest.mark.parametrize("S,F,B,maxlen", [(1, 1, 1, 70), (2, 3, 5, 70), (8, 2, 33, 70), (4, 1, 1000, 70), (3, 5, 0, 70),
                                           (3, 4, 2, 3000), (8, 2, 1, 5000)])
@pytest.mark.parametrize("width", [0, 4, 128])
def test_permute_bags(S, F, B, maxlen, width):
    from dynamicemb.input_dist import HipOps, exclusive_offsets

    rng = np.random.default_rng(S * 100 + F * 10 + B + width)
    lengths = rng.integers(0, maxlen, S * F * B).astype(np.int64)
    if lengths.size:
        lengths[rng.integers(0, lengths.size)] = 0
    n = int(lengths.sum())
    if width == 0:
        data = rng.integers(-2**62, 2**62, n).astype(np.int64)
    else:
        data = rng.standard_normal((n, width)).astype(np.float32)
    exp_len, exp = _perm_ref(S, F, B, lengths, data)
    ops = HipOps()
    l_d = torch.from_numpy(lengths).cuda()
    d_d = torch.from_numpy(data).cuda()
    out_len = ops.permute_lengths(S, F, B, l_d)
    assert np.array_equal(out_len.cpu().numpy(), exp_len)
    out = ops.permute_bags(S, F, B, exclusive_offsets(l_d), exclusive_offsets(out_len), d_d)
    assert np.array_equal(out.cpu().numpy(), exp)
    # the inverse permutation (roles of S and F swapped) restores the input
    back = ops.permute_bags(F, S, B, exclusive_offsets(out_len), exclusive_offsets(l_d), out)
    assert np.array_equal(back.cpu().numpy(), data)


@pytest.mark.parametrize("n", [16_385, 300_000, 5_000_001])
def test_single_pass_offsets_scan_repeats_cleanly(n):
    """the chained one-launch scan (decoupled look-back over the tile sums): more tiles than resident blocks, and its status
    words are clean again after every call -- three calls in a row on different data give three right answers"""
    from dynamicemb.input_dist import HipOps, TorchGlue

    rng = np.random.default_rng(n)
    hip, ref = HipOps(), TorchGlue()
    for rep in range(3):
        lengths = torch.from_numpy(rng.integers(0, 1 << (10 * rep + 4), n).astype(np.int64)).cuda()
        assert torch.equal(hip.exclusive_offsets(lengths), ref.exclusive_offsets(lengths))


@pytest.mark.parametrize("n", [0, 1, 7, 1024, 4097, 300_000])
def test_exchange_glue_kernels_match_torch(n):
    """exclusive_offsets / peer_splits / chunk_bags (one launch each) against the torch bookkeeping they replace"""
    from dynamicemb.input_dist import HipOps, TorchGlue

    rng = np.random.default_rng(n)
    hip, ref = HipOps(), TorchGlue()
    lengths = torch.from_numpy(rng.integers(0, 50, n).astype(np.int64)).cuda()
    off = hip.exclusive_offsets(lengths)
    assert torch.equal(off, ref.exclusive_offsets(lengths))
    for W in (1, 2, 8):
        per = n // W
        if per == 0:
            continue
        other = torch.from_numpy(rng.integers(0, 9, W * per).astype(np.int64)).cuda()
        roff = ref.exclusive_offsets(other)
        assert hip.peer_splits(off, roff, per, W) == ref.peer_splits(off, roff, per, W)
    for T, chunk in ((1, 64), (3, 64), (5, 7)):
        counts = torch.from_numpy(rng.integers(0, max(2, n // T + 1), T).astype(np.int64)).cuda()
        uoff = ref.exclusive_offsets(counts)
        nchunk = max(1, (int(counts.sum()) + 3 * chunk) // chunk)     # enough chunks for the longest list, as the caller sizes it
        l1, o1 = hip.chunk_bags(uoff, T, chunk, nchunk)
        l2, o2 = ref.chunk_bags(uoff, T, chunk, nchunk)
        assert torch.equal(l1, l2) and torch.equal(o1, o2)
        assert int(l1.sum()) == int(counts.sum()) and torch.equal(o1, ref.exclusive_offsets(l1))


@pytest.mark.parametrize("chunks,n", [(1, 8), (2, 1024), (8, 65536 * 16 + 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_sum_chunks(chunks, n, dtype):
    from dynamicemb.input_dist import HipOps

    x = torch.randn(chunks, n, device="cuda")
    out = HipOps().sum_chunks(x, dtype)
    ref = x[0].clone()
    for c in range(1, chunks):  # same left-to-right fp32 order as the kernel
        ref += x[c]
    assert torch.equal(out, ref.to(dtype))


@pytest.mark.parametrize("wire", [torch.bfloat16, torch.float16])
def test_sum_chunks_reads_the_wire_type(wire):
    """partial sums that crossed the fabric in a 16-bit type are summed as they arrived: fp32 accumulation, one rounding"""
    from dynamicemb.input_dist import HipOps

    x = torch.randn(8, 65536 + 4, device="cuda").to(wire)
    for dtype in (torch.float32, torch.bfloat16):
        out = HipOps().sum_chunks(x, dtype)
        ref = x[0].float()
        for c in range(1, 8):
            ref += x[c].float()
        assert torch.equal(out, ref.to(dtype))


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist

    from conftest import rendezvous_file
    dist.init_process_group("nccl", init_method=rendezvous_file(), rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def _module(pooled, F, dim, out_dtype):
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
    from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                              DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

    opts = [DynamicEmbTableOptions(dim=dim, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                   score_strategy=DynamicEmbScoreStrategy.STEP,
                                   initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG))
            for _ in range(F)]
    m = BatchedDynamicEmbeddingTablesV2(opts, pooling_mode=DynamicEmbPoolingMode.SUM if pooled else DynamicEmbPoolingMode.NONE,
                                        output_dtype=out_dtype, optimizer=EmbOptimType.SGD, learning_rate=0.25,
                                        device=torch.device("cuda", 0))
    m.train()
    return m


def test_pooled_rows_mode_one_rank_matches_unsharded(one_rank_group):
    """Rows-back pooled mode (dedup -> row exchange -> local pooling) vs the bare pooled module: the pooled
    output is bit-identical (same fp32 rows, same bag order); updated rows agree to fp32 rounding (the
    per-row gradient is summed in two stages)."""
    from dynamicemb.sharded import RowWiseShardedPooledRows, _ModuleLocal

    F, B, dim = 3, 64, 16
    rng = np.random.default_rng(11)
    torch.manual_seed(11)
    ref = _module(True, F, dim, torch.bfloat16)
    loc = _module(False, F, dim, torch.float32)
    sh = RowWiseShardedPooledRows(_ModuleLocal(loc), list(range(F)), [1000] * F, [dim] * F, combiner=0,
                                  device=torch.device("cuda", 0), out_dtype=torch.bfloat16,
                                  dist_type_per_table=["roundrobin"] * F, chunk=8)
    gsum_abs = [torch.zeros(1000, dim, dtype=torch.float64, device="cuda") for _ in range(F)]
    for step in range(3):
        lens = rng.integers(0, 9, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        keys = torch.from_numpy(rng.zipf(1.3, off[-1]).astype(np.int64) % 1000).cuda()
        off_t = torch.from_numpy(off).cuda()
        o_ref, st = ref._forward_impl(keys, off_t, train=True)
        o_sh, ctx = sh.forward(keys, off_t, True)
        if step == 0:
            assert torch.equal(o_ref, o_sh)  # identical rows -> identical pooled sums
        else:
            # (rows differ by the bound checked below; bf16 pooled outputs of ~5 rows: 1e-3 rel + 1 bf16 ulp of the sum)
            d = (o_ref.float() - o_sh.float()).abs()
            assert bool((d <= 1e-3 * o_ref.float().abs() + 2.0 ** -7 * o_ref.float().abs().clamp(min=2.0 ** -6) + 1e-3).all())
        g = (torch.randn(B, F * dim, device="cuda") * 0.1).to(torch.bfloat16)
        ref._backward_impl(st, g)
        sh.backward(ctx, g)
        # |sum of the gradients of every key| of this step, accumulated over the steps (what the bf16 rounding acts on)
        bag = torch.repeat_interleave(torch.arange(F * B, device="cuda"), off_t[1:] - off_t[:-1])
        for t in range(F):
            m = (bag // B) == t
            gs = torch.zeros(1000, dim, dtype=torch.float64, device="cuda")
            gs.index_add_(0, keys[m], g[bag[m] % B, t * dim:(t + 1) * dim].double())
            gsum_abs[t] += gs.abs()
        probe = torch.arange(0, 1000, device="cuda", dtype=torch.int64)
        for t in range(F):
            f1, r1 = ref.lookup_rows(probe, t)
            f2, r2 = loc.lookup_rows(probe, t)
            assert torch.equal(f1, f2)
            # The single-GPU path rounds the reduced gradient of a row to the gradient dtype (bf16) once per step, like
            # the reference's reduce_grads; the two-stage path keeps the fp32 sum.  So the rows may differ by lr x half
            # a bf16 unit of |sum g| per step (bf16 keeps 8 significant bits: at most 2^-8 relative), and by nothing else: the bound below is exactly that, from
            # the per-key gradient sums of the steps so far (fp64), plus an fp32 rounding floor.
            torch.testing.assert_close(f1, f2)
            bound = (0.25 * 2.0 ** -8 * gsum_abs[t] + 2e-6).float()
            assert bool(((r1 - r2).abs() <= bound).all()), float(((r1 - r2).abs() - bound).max())


@pytest.mark.parametrize("pooled", [True, False])
def test_sharded_lookup_one_rank_matches_unsharded(one_rank_group, pooled):
    """W=1: the sharded wrapper (bucketize, all-to-all, recat, output dist, backward) around a module must
    reproduce the bare module bit for bit, forward output and updated rows."""
    from dynamicemb.sharded import RowWiseShardedLookup, _ModuleLocal

    F, B, dim = 3, 17, 16
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    ref = _module(pooled, F, dim, torch.float32)
    loc = _module(pooled, F, dim, torch.float32)
    sh = RowWiseShardedLookup(_ModuleLocal(loc), F, [1000] * F, pooled=pooled, device=torch.device("cuda", 0),
                              out_dtype=torch.float32, dist_type_per_feature=["roundrobin"] * F)
    for step in range(3):
        lens = rng.integers(0, 6, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        keys = torch.from_numpy(rng.integers(0, 1000, off[-1]).astype(np.int64)).cuda()
        off_t = torch.from_numpy(off).cuda()
        o_ref, st = ref._forward_impl(keys, off_t, train=True)
        o_sh, ctx = sh.forward(keys, off_t, True)
        assert torch.equal(o_ref, o_sh)
        g = torch.randn_like(o_ref)
        ref._backward_impl(st, g)
        sh.backward(ctx, g)
        probe = torch.arange(0, 1000, device="cuda", dtype=torch.int64)
        for t in range(F):
            f1, r1 = ref.lookup_rows(probe, t)
            f2, r2 = loc.lookup_rows(probe, t)
            assert torch.equal(f1, f2) and torch.equal(r1, r2)


@pytest.mark.parametrize("pooled", [True, False])
def test_overlapped_steps_on_the_exchange_stream_match_the_plain_schedule(one_rank_group, pooled):
    """OverlappedSteps: the input dist of batch i+1 runs on the side (exchange) HIP stream under the lookup / backward of
    batch i; outputs and rows are those of the plain schedule, bit for bit"""
    from dynamicemb.sharded import OverlappedSteps, RowWiseShardedLookup, _ModuleLocal

    F, B, dim = 2, 64, 16
    rng = np.random.default_rng(9)
    mods = [_module(pooled, F, dim, torch.float32) for _ in range(2)]
    shs = [RowWiseShardedLookup(_ModuleLocal(m), F, [5000] * F, pooled=pooled, device=torch.device("cuda", 0),
                                out_dtype=torch.float32, dist_type_per_feature=["hash_roundrobin"] * F) for m in mods]
    batches = []
    for _ in range(6):
        lens = rng.integers(0, 8, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        batches.append((torch.from_numpy(rng.integers(0, 5000, off[-1]).astype(np.int64)).cuda(), torch.from_numpy(off).cuda()))
    ov = OverlappedSteps(shs[1])
    ov.prefetch(*batches[0])
    for i, (k, o) in enumerate(batches):
        o_plain, c_plain = shs[0].forward(k, o, True)
        o_ov, c_ov = ov.forward(k, o, True, batches[i + 1] if i + 1 < len(batches) else None)
        assert torch.equal(o_plain, o_ov)
        g = torch.randn_like(o_plain)
        shs[0].backward(c_plain, g)
        ov.backward(c_ov, g)
    assert shs[1]._comm is not None and shs[1]._comm != torch.cuda.current_stream()
    probe = torch.arange(0, 5000, device="cuda", dtype=torch.int64)
    for t in range(F):
        f1, r1 = mods[0].lookup_rows(probe, t)
        f2, r2 = mods[1].lookup_rows(probe, t)
        assert torch.equal(f1, f2) and torch.equal(r1, r2)


def test_bf16_wire_for_partial_sums_is_one_rounding_away(one_rank_group):
    from dynamicemb.sharded import RowWiseShardedLookup, _ModuleLocal

    F, B, dim = 2, 32, 16
    rng = np.random.default_rng(10)
    a, b = _module(True, F, dim, torch.float32), _module(True, F, dim, torch.float32)
    s32 = RowWiseShardedLookup(_ModuleLocal(a), F, [1000] * F, pooled=True, device=torch.device("cuda", 0), out_dtype=torch.float32,
                               dist_type_per_feature=["roundrobin"] * F)
    s16 = RowWiseShardedLookup(_ModuleLocal(b), F, [1000] * F, pooled=True, device=torch.device("cuda", 0), out_dtype=torch.float32,
                               dist_type_per_feature=["roundrobin"] * F, wire_dtype=torch.bfloat16)
    lens = rng.integers(0, 6, F * B)
    off = np.zeros(F * B + 1, np.int64)
    off[1:] = np.cumsum(lens)
    keys = torch.from_numpy(rng.integers(0, 1000, off[-1]).astype(np.int64)).cuda()
    o32, _ = s32.forward(keys, torch.from_numpy(off).cuda(), True)
    o16, _ = s16.forward(keys, torch.from_numpy(off).cuda(), True)
    assert torch.equal(o16, o32.bfloat16().float())      # W = 1: exactly the bf16 rounding of the fp32 sums


@pytest.mark.parametrize("pooled", [True, False])
def test_fixed_capacity_exchange_on_the_gpu_matches_the_exact_one(one_rank_group, pooled):
    """capacity_factor: padded peer slots (invalid keys get no table slot: zero rows, no update), no host read of the
    per-peer counts; same outputs and rows as the exact all-to-all-v"""
    from dynamicemb.sharded import OverlappedSteps, RowWiseShardedLookup, _ModuleLocal

    F, B, dim = 2, 64, 16
    rng = np.random.default_rng(19)
    mods = [_module(pooled, F, dim, torch.float32) for _ in range(2)]
    exact = RowWiseShardedLookup(_ModuleLocal(mods[0]), F, [5000] * F, pooled=pooled, device=torch.device("cuda", 0),
                                 out_dtype=torch.float32, dist_type_per_feature=["hash_roundrobin"] * F)
    fixed = OverlappedSteps(RowWiseShardedLookup(_ModuleLocal(mods[1]), F, [5000] * F, pooled=pooled, device=torch.device("cuda", 0),
                                                 out_dtype=torch.float32, dist_type_per_feature=["hash_roundrobin"] * F,
                                                 capacity_factor=2.0, expected_keys=F * B * 4))
    batches = []
    for _ in range(5):
        lens = rng.integers(0, 8, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        batches.append((torch.from_numpy(rng.integers(0, 5000, off[-1]).astype(np.int64)).cuda(), torch.from_numpy(off).cuda()))
    fixed.prefetch(*batches[0])
    for i, (k, o) in enumerate(batches):
        o1, c1 = exact.forward(k, o, True)
        o2, c2 = fixed.forward(k, o, True, batches[i + 1] if i + 1 < len(batches) else None)
        assert torch.equal(o1, o2)
        g = torch.randn_like(o1)
        exact.backward(c1, g)
        fixed.backward(c2, g)
    fixed.lookup.input_dist.check_overflow()
    probe = torch.arange(0, 5000, device="cuda", dtype=torch.int64)
    for t in range(F):
        f1, r1 = mods[0].lookup_rows(probe, t)
        f2, r2 = mods[1].lookup_rows(probe, t)
        assert torch.equal(f1, f2) and torch.equal(r1, r2)
        assert int(mods[1].size(t)) == int(mods[0].size(t))      # the padding keys were never stored


@pytest.mark.parametrize("pooled", [True, False])
def test_in_library_exchange_matches_the_c10d_sequence(one_rank_group, pooled, monkeypatch):
    """GPU batches go through the library's own RCCL calls (csrc/exchange.hip, one C call per stage); MI355_NATIVE_EXCHANGE=0
    keeps the c10d call sequence.  Same keys / offsets / splits out of the input dist (sync and on the exchange stream), same
    outputs, same rows after the backward -- bit for bit"""
    from dynamicemb.sharded import OverlappedSteps, RowWiseShardedLookup, _ModuleLocal

    F, B, dim = 3, 40, 16
    rng = np.random.default_rng(21)
    mods = [_module(pooled, F, dim, torch.float32) for _ in range(2)]
    monkeypatch.setenv("MI355_NATIVE_EXCHANGE", "0")
    plain = RowWiseShardedLookup(_ModuleLocal(mods[0]), F, [3000] * F, pooled=pooled, device=torch.device("cuda", 0),
                                 out_dtype=torch.float32, dist_type_per_feature=["roundrobin"] * F)
    batches = []
    for _ in range(5):
        lens = rng.integers(0, 7, F * B)
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        batches.append((torch.from_numpy(rng.integers(0, 3000, off[-1]).astype(np.int64)).cuda(), torch.from_numpy(off).cuda()))
    sk_p = plain.dist_input(*batches[0])
    assert plain._nx is None
    monkeypatch.setenv("MI355_NATIVE_EXCHANGE", "1")
    nat = RowWiseShardedLookup(_ModuleLocal(mods[1]), F, [3000] * F, pooled=pooled, device=torch.device("cuda", 0),
                               out_dtype=torch.float32, dist_type_per_feature=["roundrobin"] * F)
    sk_n = nat.dist_input(*batches[0])
    assert nat._nx is not None, "the in-library exchange must serve GPU batches by default"
    sk_a = nat.dist_input_async(*batches[0], two_phase=True).wait()
    for sk in (sk_n, sk_a):
        assert torch.equal(sk.values, sk_p.values) and torch.equal(sk.offsets, sk_p.offsets)
        assert torch.equal(sk.lengths, sk_p.lengths) and torch.equal(sk.recv_offsets, sk_p.recv_offsets)
        assert sk.send_splits == sk_p.send_splits and sk.recv_splits == sk_p.recv_splits
        assert (sk.unbucketize_permute is None) == (sk_p.unbucketize_permute is None)
        if sk_p.unbucketize_permute is not None:
            assert torch.equal(sk.unbucketize_permute, sk_p.unbucketize_permute)
    ov = OverlappedSteps(nat)
    for i, (k, o) in enumerate(batches):
        o_p, c_p = plain.forward(k, o, True)
        o_n, c_n = ov.forward(k, o, True, batches[i + 1] if i + 1 < len(batches) else None)
        assert torch.equal(o_p, o_n)
        g = torch.randn_like(o_p)
        plain.backward(c_p, g)
        ov.backward(c_n, g)
    probe = torch.arange(0, 3000, device="cuda", dtype=torch.int64)
    for t in range(F):
        f1, r1 = mods[0].lookup_rows(probe, t)
        f2, r2 = mods[1].lookup_rows(probe, t)
        assert torch.equal(f1, f2) and torch.equal(r1, r2)


@pytest.mark.parametrize("pooled", [True, False])
def test_in_library_exchange_with_empty_batches_and_empty_bags(one_rank_group, pooled):
    """a batch without a single key, then one whose bags are mostly empty, through the library's own collectives (zero-byte sends,
    zero-row receive buffers), forward and backward; outputs against the bare module"""
    from dynamicemb.sharded import OverlappedSteps, RowWiseShardedLookup, _ModuleLocal

    F, B, dim = 2, 12, 16
    rng = np.random.default_rng(77)
    ref = _module(pooled, F, dim, torch.float32)
    loc = _module(pooled, F, dim, torch.float32)
    sh = RowWiseShardedLookup(_ModuleLocal(loc), F, [500] * F, pooled=pooled, device=torch.device("cuda", 0), out_dtype=torch.float32,
                              dist_type_per_feature=["roundrobin"] * F)
    ov = OverlappedSteps(sh)
    batches = []
    for lens in (np.zeros(F * B, np.int64), (rng.random(F * B) < 0.2).astype(np.int64) * 3, np.zeros(F * B, np.int64),
                 rng.integers(0, 4, F * B)):
        off = np.zeros(F * B + 1, np.int64)
        off[1:] = np.cumsum(lens)
        batches.append((torch.from_numpy(rng.integers(0, 500, off[-1]).astype(np.int64)).cuda(), torch.from_numpy(off).cuda()))
    for i, (k, o) in enumerate(batches):
        o_ref, st = ref._forward_impl(k, o, train=True)
        o_sh, ctx = ov.forward(k, o, True, batches[i + 1] if i + 1 < len(batches) else None)
        assert sh._nx is not None
        assert o_ref.shape == o_sh.shape and torch.equal(o_ref, o_sh)
        g = torch.randn_like(o_ref)
        ref._backward_impl(st, g)
        ov.backward(ctx, g)
    probe = torch.arange(0, 500, device="cuda", dtype=torch.int64)
    for t in range(F):
        f1, r1 = ref.lookup_rows(probe, t)
        f2, r2 = loc.lookup_rows(probe, t)
        assert torch.equal(f1, f2) and torch.equal(r1, r2)


def test_the_self_check_of_the_in_library_exchange_passes_and_a_wrong_exchange_falls_back(one_rank_group, monkeypatch):
    """What a W > 1 job does on its first batch, forced here on one rank (MI355_EXCHANGE_SELFCHECK=1): the batch goes through
    BOTH exchanges and the output collectives are compared on small blocks; then the same with a corrupted in-library
    all-to-all-v: the lookup logs, aborts its communicators and serves every batch through the c10d sequence -- same results."""
    from dynamicemb import native_exchange as ne
    from dynamicemb.sharded import RowWiseShardedLookup, _ModuleLocal

    F, B, dim = 2, 24, 16
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 6, F * B)
    off = np.zeros(F * B + 1, np.int64)
    off[1:] = np.cumsum(lens)
    keys = torch.from_numpy(rng.integers(0, 2000, off[-1]).astype(np.int64)).cuda()
    offs = torch.from_numpy(off).cuda()
    monkeypatch.setenv("MI355_EXCHANGE_SELFCHECK", "1")
    mods = [_module(True, F, dim, torch.float32) for _ in range(3)]
    mk = lambda m: RowWiseShardedLookup(_ModuleLocal(m), F, [2000] * F, pooled=True, device=torch.device("cuda", 0),
                                        out_dtype=torch.float32, dist_type_per_feature=["roundrobin"] * F)
    good = mk(mods[0])
    o_good, _ = good.forward(keys, offs, True)
    assert good.exchange == "native" and good.exchange_selfchecked

    real = ne.NativeExchange.alltoallv_rows

    def broken(self, send, sc, rc):
        out = real(self, send, sc, rc)
        out[0, 0] += 1.0
        return out

    monkeypatch.setattr(ne.NativeExchange, "alltoallv_rows", broken)
    bad = mk(mods[1])
    o_bad, _ = bad.forward(keys, offs, True)
    assert bad.exchange == "c10d" and bad._nx is None and bad._nx_off
    assert torch.equal(o_good, o_bad)
    o_again, _ = bad.forward(keys, offs, True)         # stays on the c10d sequence
    assert torch.equal(o_good, o_again)

    # a librccl.so that cannot be bound: creation fails on "every" rank, c10d from the first batch
    monkeypatch.setattr(ne.NativeExchange, "alltoallv_rows", real)
    monkeypatch.setattr(ne.NativeExchange, "_bind", lambda self, pg: (_ for _ in ()).throw(OSError("no librccl.so")))
    none = mk(mods[2])
    o_none, _ = none.forward(keys, offs, True)
    assert none.exchange == "c10d" and torch.equal(o_good, o_none)


def test_input_dists_in_flight_are_bounded_by_the_ticket_ring(one_rank_group):
    """eight input dists may be in flight; a ninth begin before the oldest finish is refused instead of overwriting the pinned
    counts and events of a live ticket"""
    from mi355_native import NativeError
    from dynamicemb.sharded import RowWiseShardedLookup, _ModuleLocal

    F, B = 1, 16
    m = _module(True, F, 16, torch.float32)
    sh = RowWiseShardedLookup(_ModuleLocal(m), F, [500], pooled=True, device=torch.device("cuda", 0), out_dtype=torch.float32,
                              dist_type_per_feature=["roundrobin"])
    keys = torch.arange(0, 2 * B, dtype=torch.int64, device="cuda")
    offs = torch.arange(0, 2 * B + 1, 2, dtype=torch.int64, device="cuda")
    pend = [sh.dist_input_async(keys, offs, two_phase=True) for _ in range(8)]
    with pytest.raises(NativeError, match="ticket"):
        sh.dist_input_async(keys, offs, two_phase=True)
    first = pend[0].wait()
    assert torch.equal(first.values, keys)
    pend.append(sh.dist_input_async(keys, offs, two_phase=True))      # a slot is free again
    for p in pend[1:]:
        assert torch.equal(p.wait().values, keys)


def _two_rank_worker(rank, world, store, pooled, fail):
    """one process per GPU: the in-library exchange against the c10d sequence at W = 2, forward and backward, overlapped
    schedule; any assertion ends the rank with a traceback (spawn turns it into a failure of the test)"""
    import torch.distributed as dist

    os.environ.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", init_method=store, rank=rank, world_size=world, device_id=dev)
    try:
        from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
        from dynamicemb.dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                                  DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)
        from dynamicemb.sharded import OverlappedSteps, RowWiseShardedLookup, _ModuleLocal

        F, B, dim = 3, 40, 16

        def module():
            opts = [DynamicEmbTableOptions(dim=dim, max_capacity=4096, index_type=torch.int64, embedding_dtype=torch.float32,
                                           score_strategy=DynamicEmbScoreStrategy.STEP,
                                           initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.DEBUG))
                    for _ in range(F)]
            m = BatchedDynamicEmbeddingTablesV2(opts, pooling_mode=DynamicEmbPoolingMode.SUM if pooled else DynamicEmbPoolingMode.NONE,
                                                output_dtype=torch.float32, optimizer=EmbOptimType.SGD, learning_rate=0.25, device=dev)
            m.train()
            return m

        mods = [module(), module()]
        rng = np.random.default_rng(100 + rank)
        batches = []
        for _ in range(5):
            lens = rng.integers(0, 7, F * B)
            off = np.zeros(F * B + 1, np.int64)
            off[1:] = np.cumsum(lens)
            batches.append((torch.from_numpy(rng.integers(0, 3000, off[-1]).astype(np.int64)).to(dev), torch.from_numpy(off).to(dev)))
        os.environ["MI355_NATIVE_EXCHANGE"] = "0"
        plain = RowWiseShardedLookup(_ModuleLocal(mods[0]), F, [3000] * F, pooled=pooled, device=dev, out_dtype=torch.float32,
                                     dist_type_per_feature=["roundrobin"] * F)
        sk_p = plain.dist_input(*batches[0])
        assert plain.exchange == "c10d"
        os.environ["MI355_NATIVE_EXCHANGE"] = "1"
        if fail:     # rank 1 cannot bind RCCL: BOTH ranks must land on the c10d sequence (nobody is left inside a collective)
            from dynamicemb import native_exchange as ne

            if rank == 1:
                ne.NativeExchange._bind = lambda self, pg: (_ for _ in ()).throw(OSError("no librccl.so on this rank"))
        nat = RowWiseShardedLookup(_ModuleLocal(mods[1]), F, [3000] * F, pooled=pooled, device=dev, out_dtype=torch.float32,
                                   dist_type_per_feature=["roundrobin"] * F)
        sk_n = nat.dist_input(*batches[0])
        assert nat.exchange == ("c10d" if fail else "native"), nat.exchange
        assert fail or nat.exchange_selfchecked
        assert torch.equal(sk_n.values, sk_p.values) and torch.equal(sk_n.offsets, sk_p.offsets)
        assert sk_n.send_splits == sk_p.send_splits and sk_n.recv_splits == sk_p.recv_splits
        ov = OverlappedSteps(nat)
        for i, (k, o) in enumerate(batches):
            o_p, c_p = plain.forward(k, o, True)
            o_n, c_n = ov.forward(k, o, True, batches[i + 1] if i + 1 < len(batches) else None)
            assert torch.equal(o_p, o_n)
            g = torch.randn_like(o_p)
            plain.backward(c_p, g)
            ov.backward(c_n, g)
        probe = torch.arange(0, 3000, device=dev, dtype=torch.int64)
        for t in range(F):
            f1, r1 = mods[0].lookup_rows(probe, t)
            f2, r2 = mods[1].lookup_rows(probe, t)
            assert torch.equal(f1, f2) and torch.equal(r1, r2)
        torch.cuda.synchronize()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail", [False, True])
@pytest.mark.parametrize("pooled", [True, False])
def test_two_ranks_in_library_exchange_matches_the_c10d_sequence(pooled, fail):
    """The W = 2 run of `test_in_library_exchange_matches_the_c10d_sequence` -- one process per GPU over RCCL.  Skipped on a
    one-GPU box (two ranks cannot share a device under RCCL); on any box with two GPUs it is part of `-m gpu`."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per device)")
    import torch.multiprocessing as mp

    from conftest import rendezvous_file
    mp.spawn(_two_rank_worker, args=(2, rendezvous_file(), pooled, fail), nprocs=2, join=True)
