"""World-size-2 (and 3) gloo tests of the row-wise sharded lookup's routing / collective logic on CPU.

The product code under test is dynamicemb/input_dist.py + dynamicemb/sharded.py (bucketize -> all-to-all
lengths/keys -> recat -> local lookup -> output dist, and the backward).  There is no GPU here, so the
element work is injected: a numpy `ops` backend built on the oracle's block_bucketize, and a dict-backed
local table.  Expected results come from ONE process doing the global batch on one dict table.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "recsys-examples_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

D = 8
LR = 0.5


def init_row(key: int, table: int = 0) -> np.ndarray:
    return ((np.arange(D, dtype=np.float32) + 1.0 + table) * np.float32((key % 97) + 1) * np.float32(0.01)).astype(np.float32)


def _glue_base():
    from dynamicemb.input_dist import TorchGlue

    return TorchGlue


class NumpyOps(_glue_base()):
    """CPU stand-in for HipOps (tests only): same contracts, oracle/numpy inside; the index bookkeeping
    (offsets, peer splits, pseudo-bags, index composition) is the product's own torch code (TorchGlue)."""

    def bucketize(self, offsets, values, block_sizes, world, sequence, dist_types):
        from oracle import oracle as orc

        off = offsets.numpy().astype(np.int64)
        F = block_sizes.numel()
        B = (off.size - 1) // F
        dts = set(dist_types.tolist())
        assert len(dts) == 1
        nl, _, ni, perm = orc.block_bucketize(off, values.numpy(), world, B, block_sizes.numpy(), dts.pop())
        no = np.zeros(nl.size + 1, np.int64)
        no[1:] = np.cumsum(nl)
        return (torch.from_numpy(nl.astype(np.int64)), torch.from_numpy(no), torch.from_numpy(ni.astype(np.int64)),
                torch.from_numpy(perm) if sequence else None)

    def permute_lengths(self, S, F, B, lengths):
        return lengths.view(S, F, B).permute(1, 0, 2).contiguous().view(-1)

    def permute_bags(self, S, F, B, in_offsets, out_offsets, data):
        out = torch.empty_like(data)
        io, oo = in_offsets.tolist(), out_offsets.tolist()
        for f in range(F):
            for s in range(S):
                for b in range(B):
                    i = s * F * B + f * B + b
                    o = f * S * B + s * B + b
                    n = io[i + 1] - io[i]
                    assert oo[o + 1] - oo[o] == n
                    out[oo[o]:oo[o] + n] = data[io[i]:io[i] + n]
        return out

    def sum_chunks(self, x, out_dtype):
        return x.float().sum(0).to(out_dtype)

    def gather_rows(self, src, index):
        return src[index]

    def unique(self, keys, offsets, feature_offsets):
        from oracle import oracle as orc

        rng = orc.get_table_range(offsets.numpy(), feature_offsets.numpy())
        uk, rev, uoff, _ = orc.segmented_unique(keys.numpy(), rng)
        padded = np.full(keys.numel(), -7, np.int64)  # padded like the device buffer: only [:Nu] is valid
        padded[:uk.size] = uk.astype(np.int64)
        return torch.from_numpy(padded), torch.from_numpy(rev), torch.from_numpy(uoff), None

    def pool(self, rows, reverse, offsets, batch_size, combiner, total_D, D_offsets, max_D, out_dtype):
        from oracle import oracle as orc

        out = orc.gather_pooled(rows.numpy(), reverse.numpy(), offsets.numpy(), batch_size, combiner)
        return torch.from_numpy(out).to(out_dtype)

    def reduce_grads(self, reverse, grads, num_unique, batch_size, dim, offsets, D_offsets, combiner, aux=None):
        from oracle import oracle as orc

        return torch.from_numpy(orc.reduce_grads(reverse.numpy(), grads.numpy(), num_unique, batch_size,
                                                 offsets.numpy(), None, combiner))


class DictLocal:
    """A dict-backed embedding shard, one table per feature: rows[(table, key)], insert-on-miss with init_row,
    SUM pooling or sequence rows, SGD."""

    def __init__(self, num_features, pooled, key_offset=0):
        # "continuous" routing re-bases keys to the shard (new_idx = idx - rank*block); the other modes send
        # the key unchanged (sparse_block_bucketize_features.cu:330-341)
        self.key_offset = key_offset
        self.rows = {}
        self.F = num_features
        self.pooled = pooled

    INVALID = -1   # the padding key of the fixed-capacity exchange: no row, no update (as the table kernels treat it)

    def _tagged(self, values, off):
        nb = len(off) - 1
        B = nb // self.F
        keys = []
        for bag in range(nb):
            t = bag // B if B else 0
            keys += [(t, k + self.key_offset) for k in values[off[bag]:off[bag + 1]]]
        return keys, B

    def forward(self, values, offsets, train):
        off = offsets.tolist()
        keys, B = self._tagged(values.tolist(), off)
        zero = np.zeros(D, np.float32)
        for tk in keys:
            if tk[1] - self.key_offset != self.INVALID and tk not in self.rows:
                self.rows[tk] = init_row(tk[1], tk[0])
        if self.pooled:
            out = np.zeros((B, self.F * D), np.float32)
            for f in range(self.F):
                for b in range(B):
                    bag = f * B + b
                    for j in range(off[bag], off[bag + 1]):
                        out[b, f * D:(f + 1) * D] += self.rows.get(keys[j], zero)
        else:
            out = np.stack([self.rows.get(k, zero) for k in keys]) if keys else np.zeros((0, D), np.float32)
        return torch.from_numpy(out), (keys, off, B)

    def backward(self, ctx, grads):
        keys, off, B = ctx
        g = grads.float().numpy()
        acc = {}
        if self.pooled:
            for f in range(self.F):
                for b in range(B):
                    bag = f * B + b
                    for j in range(off[bag], off[bag + 1]):
                        acc[keys[j]] = acc.get(keys[j], 0) + g[b, f * D:(f + 1) * D]
        else:
            for j, k in enumerate(keys):
                acc[k] = acc.get(k, 0) + g[j]
        for k, v in acc.items():
            if k in self.rows:
                self.rows[k] = (self.rows[k] - np.float32(LR) * v).astype(np.float32)


def make_batch(rank, F, B, seed, max_len=5, key_space=200):
    rng = np.random.default_rng(seed * 100 + rank)
    lens = rng.integers(0, max_len + 1, F * B)
    if rank == 0:
        lens[0] = 0  # an empty bag
    off = np.zeros(F * B + 1, np.int64)
    off[1:] = np.cumsum(lens)
    keys = rng.integers(0, key_space, off[-1]).astype(np.int64)
    return torch.from_numpy(keys), torch.from_numpy(off)


def grads_for(rank, shape, seed):
    rng = np.random.default_rng(seed * 7 + rank + 1000)
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def _worker(rank, W, store, F, B, pooled, dist_type, steps, q):
    dist.init_process_group("gloo", init_method=store, rank=rank, world_size=W)
    try:
        from dynamicemb.sharded import RowWiseShardedLookup, RowWiseShardedPooledRows

        blk = (200 + W - 1) // W
        koff = rank * blk if dist_type == "continuous" else 0
        if pooled == "rows":
            # two features share table 0, the third has its own: T = 2 when F == 3, else one table per feature
            ftm = [0, 0, 1] if F == 3 else list(range(F))
            T = max(ftm) + 1
            local = DictLocal(T, False, key_offset=koff)
            sh = RowWiseShardedPooledRows(local, ftm, [200] * T, [D] * T, combiner=0, device="cpu",
                                          out_dtype=torch.float32, dist_type_per_table=[dist_type] * T, ops=NumpyOps(),
                                          chunk=4)
        else:
            local = DictLocal(F, pooled, key_offset=koff)
            cf = float(os.environ["TEST_CAPACITY_FACTOR"]) if os.environ.get("TEST_CAPACITY_FACTOR") else None
            odt = torch.bfloat16 if os.environ.get("TEST_OUT_DTYPE") == "bf16" else torch.float32
            sh = RowWiseShardedLookup(local, F, [200] * F, pooled=pooled, device="cpu", out_dtype=odt,
                                      wire_dtype=os.environ.get("TEST_WIRE_DTYPE") or None,
                                      dist_type_per_feature=[dist_type] * F, ops=NumpyOps(), capacity_factor=cf,
                                      expected_keys=F * B * 5 if cf else None)
        outs = []
        if pooled in (True, False) and os.environ.get("TEST_OVERLAPPED") == "1":
            from dynamicemb.sharded import OverlappedSteps

            # batch i+1's key exchange is issued before batch i's output dist and backward (the overlapped schedule)
            st = OverlappedSteps(sh)
            batches = [make_batch(rank, F, B, step) for step in range(steps)]
            st.prefetch(*batches[0])
            for step in range(steps):
                out, ctx = st.forward(*batches[step], True, batches[step + 1] if step + 1 < steps else None)
                outs.append(out.float().numpy().copy())
                st.backward(ctx, grads_for(rank, tuple(out.shape), step))
        else:
            for step in range(steps):
                keys, off = make_batch(rank, F, B, step)
                out, ctx = sh.forward(keys, off, True)
                outs.append(out.float().numpy().copy())
                sh.backward(ctx, grads_for(rank, tuple(out.shape), step))
        q.put((rank, outs, dict(local.rows)))
    finally:
        dist.destroy_process_group()


def _ftm(F, pooled):
    return ([0, 0, 1] if F == 3 else list(range(F))) if pooled == "rows" else list(range(F))


def _expected(W, F, B, pooled, steps):
    """One process, one set of tables, the global batch (rank r's samples are batch rows r*B..(r+1)*B)."""
    ftm = _ftm(F, pooled)
    rows = {}
    outs = [[] for _ in range(W)]
    for step in range(steps):
        batches = [make_batch(r, F, B, step) for r in range(W)]
        tagged = []
        for r in range(W):
            keys, off = batches[r][0].tolist(), batches[r][1].tolist()
            tk = []
            for bag in range(F * B):
                tk += [(ftm[bag // B], k) for k in keys[off[bag]:off[bag + 1]]]
            tagged.append(tk)
            for x in tk:
                rows.setdefault(x, init_row(x[1], x[0]))
        acc = {}
        for r in range(W):
            keys, off = tagged[r], batches[r][1].tolist()
            if pooled:
                out = np.zeros((B, F * D), np.float32)
                for f in range(F):
                    for b in range(B):
                        for j in range(off[f * B + b], off[f * B + b + 1]):
                            out[b, f * D:(f + 1) * D] += rows[keys[j]]
            else:
                out = np.stack([rows[k] for k in keys]) if keys else np.zeros((0, D), np.float32)
            outs[r].append(out)
            g = grads_for(r, out.shape, step).numpy()
            if pooled:
                for f in range(F):
                    for b in range(B):
                        for j in range(off[f * B + b], off[f * B + b + 1]):
                            acc[keys[j]] = acc.get(keys[j], 0) + g[b, f * D:(f + 1) * D]
            else:
                for j, k in enumerate(keys):
                    acc[k] = acc.get(k, 0) + g[j]
        for k, v in acc.items():
            rows[k] = (rows[k] - np.float32(LR) * v).astype(np.float32)
    return outs, rows


def _owner(key, W, dist_type):
    if dist_type == "roundrobin":
        return key % W
    if dist_type == "continuous":
        return key // ((200 + W - 1) // W)
    from oracle import oracle as orc

    return orc.fmix64(key) % W


@pytest.mark.parametrize("W,F,B,pooled,dist_type", [
    (2, 2, 5, True, "roundrobin"),
    (2, 3, 4, False, "roundrobin"),
    (2, 1, 6, True, "hash_roundrobin"),
    (2, 2, 3, False, "continuous"),
    (3, 2, 4, True, "roundrobin"),
    (2, 3, 6, "rows", "roundrobin"),
    (2, 2, 5, "rows", "hash_roundrobin"),
    (3, 3, 4, "rows", "continuous"),
])
@pytest.mark.parametrize("overlapped", [False, True])
def test_rowwise_sharded_matches_single_process(W, F, B, pooled, dist_type, overlapped, monkeypatch):
    """overlapped: the key exchange of batch i+1 runs ahead of batch i's output dist / backward (OverlappedSteps) -- the
    results must be those of the plain schedule, i.e. of one process doing the global batch"""
    if overlapped and pooled == "rows":
        pytest.skip("the rows-back pooled mode has its own two-level schedule")
    monkeypatch.setenv("TEST_OVERLAPPED", "1" if overlapped else "0")
    _run_and_compare(W, F, B, pooled, dist_type)


def _run_and_compare(W, F, B, pooled, dist_type, out_rtol=1e-5, out_atol=1e-6):
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import rendezvous_file
    store = rendezvous_file()
    procs = [ctx.Process(target=_worker, args=(r, W, store, F, B, pooled, dist_type, steps, q)) for r in range(W)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(W):
        rank, outs, rows = q.get(timeout=600)
        got[rank] = (outs, rows)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_outs, exp_rows = _expected(W, F, B, pooled, steps)
    seen = {}
    for r in range(W):
        outs, rows = got[r]
        for step in range(steps):
            np.testing.assert_allclose(outs[step], exp_outs[r][step], rtol=out_rtol, atol=out_atol,
                                       err_msg=f"rank {r} step {step}")
        for k, v in rows.items():
            assert _owner(k[1], W, dist_type) == r, f"key {k} landed on rank {r}"
            assert k not in seen
            seen[k] = v
    assert set(seen) == set(exp_rows)
    for k in exp_rows:
        np.testing.assert_allclose(seen[k], exp_rows[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("W,F,B,pooled,dist_type", [(2, 2, 5, True, "roundrobin"), (2, 3, 4, False, "hash_roundrobin"),
                                                    (3, 2, 4, False, "roundrobin")])
def test_fixed_capacity_exchange_matches_single_process(W, F, B, pooled, dist_type, monkeypatch):
    """capacity_factor: every peer slot of the key exchange has a fixed size (padded with invalid keys), so no per-peer
    count is read back; results are those of the exact exchange"""
    monkeypatch.setenv("TEST_CAPACITY_FACTOR", "3.0")
    monkeypatch.setenv("TEST_OVERLAPPED", "1")
    _run_and_compare(W, F, B, pooled, dist_type)


@pytest.mark.parametrize("W", [2, 3])
def test_bf16_outputs_take_the_bf16_wire(W, monkeypatch):
    """bf16 pooled outputs (what bench.py --gpus N asks for): with wire_dtype "auto" (the benchmark helper's default; the
    TorchRec-facing collection exchanges fp32 unless fused_params opt in) the partial sums cross the fabric in bf16,
    are summed in fp32 and rounded once more -- every output within (W + 1) half-ulps of the bf16
    value of the exact sum; the table rows, updated from fp32 gradients, are exactly those of the fp32 run."""
    monkeypatch.setenv("TEST_OUT_DTYPE", "bf16")
    monkeypatch.setenv("TEST_WIRE_DTYPE", "auto")
    monkeypatch.setenv("TEST_OVERLAPPED", "1")
    _run_and_compare(W, 2, 5, True, "roundrobin", out_rtol=(W + 1) * 2.0 ** -9, out_atol=(W + 1) * 2.0 ** -9 * 8.0)


def test_fixed_capacity_overflow_is_reported():
    """one rank, capacity far below the batch: the overflow flag turns into an error at the next exchange"""
    from dynamicemb.input_dist import RwSparseFeaturesDist

    from conftest import rendezvous_file
    dist.init_process_group("gloo", init_method=rendezvous_file(), rank=0, world_size=1)
    try:
        d = RwSparseFeaturesDist(dist.group.WORLD, 1, [200], "cpu", is_sequence=True, dist_type_per_feature=["roundrobin"],
                                 ops=NumpyOps(), capacity_factor=0.25, expected_keys=8)   # slot of 8 keys
        keys, off = make_batch(0, 1, 40, 0)        # ~100 keys into a slot of 8
        assert keys.numel() > 8
        d(off[1:] - off[:-1], keys, offsets=off)
        with pytest.raises(RuntimeError, match="overflowed"):
            d(off[1:] - off[:-1], keys, offsets=off)
    finally:
        dist.destroy_process_group()


def test_per_link_bytes_decide_the_pooled_dist_mode():
    """xGMI is a full mesh: the dist that puts fewer bytes on ONE peer link wins.  At C2 the fp32 wire
    keeps the reference's partial-sum dist up to W = 4 and switches to the rows-back dist at W = 8; the bf16 wire (the default
    whenever the caller wants bf16 outputs) halves the forward block and keeps partial sums at every W of one node."""
    from dynamicemb.sharded import ShardedPooledLookup as S

    Nt, B, Dm = 360_000, 65_536, 128
    assert [S.choose_mode(w, Nt, B, Dm, torch.bfloat16, torch.float32) for w in (2, 4, 8)] == ["partial", "partial", "rows"]
    assert [S.choose_mode(w, Nt, B, Dm, torch.bfloat16) for w in (2, 4, 8)] == ["partial"] * 3
    assert S.choose_mode(8, Nt, B, Dm, torch.float32) == "rows"          # fp32 outputs: fp32 wire
    assert S.choose_mode(8, None, None, Dm) == "partial"                  # nothing known about the batch: the reference's dist
