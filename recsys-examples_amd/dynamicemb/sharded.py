"""Row-wise (model-parallel) sharded DynamicEmb lookup: one process per GPU, one exchange each way.

Mirror of what TorchRec's RwPooledEmbeddingSharding / RwSequenceEmbeddingSharding do around the
reference's compute kernel (planner/rw_sharding.py:85-158 sequence, :191-261 pooled; output dists are
third-party TorchRec, restated from SURVEY.md 8(e)):

  input dist   bucketize -> all-to-all lengths -> all-to-all-v keys -> recat         (input_dist.py)
  lookup       the full single-GPU path on the received keys (BatchedDynamicEmbeddingTablesV2)
  output dist  pooled:   partial sums for the GLOBAL batch [W*B, total_D] fp32 -> all-to-all of the W
                         [B, total_D] blocks + one local sum kernel.  The reference uses a ring
                         reduce-scatter; on MI355X every GPU pair owns an xGMI link, so the all-to-all
                         moves (W-1)/W of the data over 7 links in parallel where a ring is bound by one.
               sequence: all-to-all-v of rows back + unbucketize_permute gather
  backward     pooled: all-gather of output grads; sequence: all-to-all-v of row grads; then the local
               fused backward (reduce + optimizer) on each shard.  No cross-GPU atomics anywhere: the
               hash tables are per-rank and independent.

The local lookup and all element work are injected (`local`, `ops`) so the routing/collective logic runs
unchanged over gloo on CPU in the tests, with an oracle-backed local lookup injected by the test.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .input_dist import DIST_TYPES, HipOps, RwSparseFeaturesDist, ShardedKeys


def current_torch_stream():
    from mi355_native import current_torch_stream as f     # (GPU paths only; the gloo tests never get here)
    return f()


def on_stream(s):
    from mi355_native import on_stream as c
    return c(s)


@dataclass
class _ShardedCtx:
    keys: ShardedKeys
    local_ctx: object
    n_local: int


class PendingKeys:
    """an input dist in flight on the exchange stream; wait() orders the current stream behind it and hands the keys over.
    Two-phase form: `state` is what RwSparseFeaturesDist.forward(two_phase=True) left; finish() -- called when the caller
    has queued its other work -- reads the per-peer key counts and sends the keys (still on the exchange stream)."""

    def __init__(self, sk, event=None, dist=None, state=None, comm=None, consumer=None, native=None):
        self._sk, self._event = sk, event
        self._dist, self._state, self._comm, self._consumer = dist, state, comm, consumer
        self._native = native        # the in-library exchange (native_exchange.py): state is its input_begin() state

    def try_finish(self) -> bool:
        """finish() if that does not make the host wait (in-library exchange: the key counts have been written); False: not yet"""
        if self._state is None:
            return True
        if self._native is not None and not self._native.counts_ready(self._state):
            return False
        self.finish()
        return True

    def finish(self):
        if self._state is None:
            return
        state, self._state = self._state, None
        if self._native is not None:
            self._sk = self._native.input_finish(state)
            return
        if self._comm is None:
            self._sk = self._dist.finish(state)
            return
        with on_stream(self._comm):
            sk = self._dist.finish(state)
            ev = torch.cuda.Event()
            ev.record(self._comm)
        for t in (sk.lengths, sk.offsets, sk.values, sk.recv_offsets, sk.unbucketize_permute):
            if t is not None and self._consumer is not None:
                t.record_stream(self._consumer)      # allocated on the exchange stream, consumed on the caller's
        self._sk, self._event = sk, ev

    def wait(self):
        self.finish()
        if self._native is not None:
            tk = getattr(self._sk, "_ticket", None)
            if tk is not None:           # the keys travelled on the exchange stream: the caller's stream waits for them
                self._native.wait_keys(tk)
                self._sk._ticket = None
            return self._sk
        if self._event is not None:
            current_torch_stream().wait_event(self._event)
            self._event = None
        return self._sk


class RowWiseShardedLookup:
    """`local` must provide
         forward(values, offsets, train) -> (out, ctx)   out: pooled [W*B, total_D] fp32 or rows [n, D]
         backward(ctx, grads) -> None
       (BatchedDynamicEmbeddingTablesV2._forward_impl/_backward_impl have exactly this shape)."""

    def __init__(self, local, num_features: int, feature_hash_sizes: List[int], pooled: bool, pg=None,
                 device=None, out_dtype=torch.float32, dist_type_per_feature: Optional[Sequence[str]] = None,
                 ops=None, wire_dtype=None, capacity_factor: Optional[float] = None,
                 expected_keys: Optional[int] = None):
        """wire_dtype (pooled): element type of the partial sums on the fabric.  None / torch.float32 (DEFAULT, what TorchRec
        and the reference exchange unless a qcomm codec is configured) = fp32: the sums of the shards are added in fp32, one
        rounding at the end, bit-identical to the single-GPU sum order for an fp32 output.  torch.bfloat16 (opt-in; the
        TorchRec-facing collection takes it from fused_params["wire_dtype"]) halves the bytes per xGMI link at the price of
        one more rounding per shard -- what TorchRec's qcomm codec does for its reduce-scatter.  "auto" = bf16 when the
        caller asked for bf16 OUTPUT (the result is rounded to bf16 anyway, the extra error is at most half a bf16 ulp of
        each shard's partial sum: bounded in tests/test_sharded_gpu.py; at W = 1 bit-identical), fp32 otherwise: the
        default of the benchmark helper ShardedPooledLookup only."""
        if wire_dtype == "auto":
            wire_dtype = torch.bfloat16 if (pooled and out_dtype == torch.bfloat16) else None
        self.wire_dtype = None if wire_dtype == torch.float32 else wire_dtype
        self._comm = None
        self.pg = pg if pg is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.pg)
        self.rank = dist.get_rank(self.pg)
        self.local = local
        self.pooled = pooled
        self.out_dtype = out_dtype
        self.ops = ops or HipOps()
        self.input_dist = RwSparseFeaturesDist(self.pg, num_features, feature_hash_sizes, device,
                                               is_sequence=not pooled, dist_type_per_feature=dist_type_per_feature,
                                               ops=self.ops, capacity_factor=capacity_factor, expected_keys=expected_keys)
        self.fixed_capacity = capacity_factor is not None
        self._nx = None               # the in-library exchange (GPU batches; created at the first one)
        self._nx_off = False
        self.exchange_selfchecked = False

    def _native_for(self, t: torch.Tensor, offsets: Optional[torch.Tensor] = None):
        """the library's own RCCL exchange (native_exchange.py) for GPU batches of the exact exchange; None: the c10d sequence.
        Never raises and never leaves a rank alone: creation agrees across ranks (NativeExchange.create), and at W > 1 the
        first batch goes through BOTH exchanges once (`_selftest_native`) before the in-library one is trusted."""
        if self._nx is not None:
            if self._nx.dead:
                self._nx, self._nx_off = None, True
                return None
            return self._nx if t.is_cuda else None
        if self._nx_off or self.fixed_capacity:
            return None
        from .native_exchange import NativeExchange, all_ranks_ok, native_exchange_wanted

        if not native_exchange_wanted(self.pg, t):
            self._nx_off = t.is_cuda      # (a CPU batch decides nothing: the gloo tests never get a communicator)
            return None
        nx = NativeExchange.create(self.pg, t.device)
        if nx is None:
            self._nx_off = True
            return None
        if (self.world > 1 or os.environ.get("MI355_EXCHANGE_SELFCHECK", "") == "1") and offsets is not None:
            why = None
            try:
                self._selftest_native(nx, t, offsets)
            except Exception as e:      # noqa: BLE001 -- a wrong or hung exchange must not end the run: c10d takes over
                why = e
            if not all_ranks_ok(self.pg, why is None, t.device):
                import logging

                logging.getLogger("dynamicemb.native_exchange").warning(
                    "in-library RCCL exchange failed its self-check (%s): using the c10d sequence", why or "another rank failed")
                nx.abort()
                self._nx_off = True
                return None
        self._nx = nx
        self.exchange_selfchecked = self.world > 1 or os.environ.get("MI355_EXCHANGE_SELFCHECK", "") == "1"
        return self._nx

    @property
    def exchange(self) -> str:
        """which exchange moves this lookup's GPU batches: "native" (csrc/exchange.hip) or "c10d" (torch.distributed calls)"""
        return "native" if self._nx is not None and not self._nx.dead else "c10d"

    def _selftest_native(self, nx, values, offsets) -> None:
        """One-time check of the in-library exchange against the c10d sequence, every wait bounded (30 s):
        the input dist of THIS batch through both (received keys / offsets / lengths / splits must be equal), then the three
        output collectives on small rank-dependent blocks.  Raises on any difference or timeout."""
        from .native_exchange import bounded_wait

        def fence(what):
            ev = torch.cuda.Event()
            ev.record(current_torch_stream())
            if not bounded_wait(ev.query):
                raise TimeoutError(f"in-library exchange: {what} did not complete")

        W, dev = self.world, values.device
        # --- input dist of the real batch
        state = self._native_begin(nx, values, offsets, None)
        if not bounded_wait(lambda: nx.counts_ready(state)):
            raise TimeoutError("in-library exchange: lengths all-to-all did not complete")
        sk_n = nx.input_finish(state)
        if not bounded_wait(lambda: nx.keys_ready(sk_n._last_ticket)):
            raise TimeoutError("in-library exchange: key all-to-all did not complete")
        sk_c = self.input_dist(None, values, False, offsets=offsets)
        same = (sk_n.send_splits == sk_c.send_splits and sk_n.recv_splits == sk_c.recv_splits
                and torch.equal(sk_n.values, sk_c.values) and torch.equal(sk_n.offsets, sk_c.offsets)
                and torch.equal(sk_n.lengths, sk_c.lengths) and torch.equal(sk_n.recv_offsets, sk_c.recv_offsets)
                and (sk_n.unbucketize_permute is None) == (sk_c.unbucketize_permute is None)
                and (sk_c.unbucketize_permute is None or torch.equal(sk_n.unbucketize_permute, sk_c.unbucketize_permute)))
        if not same:
            raise RuntimeError("in-library input dist differs from the c10d sequence")
        # --- output collectives on small blocks that differ per rank and per destination
        g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        Bt, Dt = 8, 16
        for wire in (torch.float32, torch.bfloat16):
            send = torch.randn(W * Bt, Dt, generator=g).to(dev).to(wire)
            out_n = nx.output_pooled(send, torch.float32)
            fence("partial-sum all-to-all")
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.pg)
            out_c = self.ops.sum_chunks(recv.view(W, Bt * Dt), torch.float32).view(Bt, Dt)
            if not torch.equal(out_n, out_c):
                raise RuntimeError("in-library pooled output dist differs from the c10d sequence")
        blk = torch.randn(Bt, Dt, generator=g).to(dev)
        ga_n = nx.allgather(blk)
        fence("all-gather")
        ga_c = torch.empty(W * Bt, Dt, dtype=blk.dtype, device=dev)
        dist.all_gather_into_tensor(ga_c, blk, group=self.pg)
        if not torch.equal(ga_n, ga_c):
            raise RuntimeError("in-library all-gather differs from the c10d one")
        sc = [1 + (self.rank + p) % 3 for p in range(W)]          # rows this rank sends to peer p
        rc = [1 + (p + self.rank) % 3 for p in range(W)]          # ... and receives from peer p (the peer's sc[self.rank])
        rows = torch.randn(sum(sc), Dt, generator=g).to(dev)
        rv_n = nx.alltoallv_rows(rows, sc, rc)
        fence("row all-to-all-v")
        rv_c = torch.empty(sum(rc), Dt, dtype=rows.dtype, device=dev)
        dist.all_to_all_single(rv_c, rows, rc, sc, group=self.pg)
        if not torch.equal(rv_n, rv_c):
            raise RuntimeError("in-library row all-to-all-v differs from the c10d one")

    def _native_begin(self, nx, values, offsets, side):
        d = self.input_dist
        if d._dist_codes is None or d._dist_codes.device != values.device:
            d._dist_codes = torch.tensor([DIST_TYPES[x] for x in d._dist_type_per_feature], dtype=torch.int32,
                                         device=values.device)
            d._block_sizes = d._block_sizes.to(values.device)
        return nx.input_begin(d._num_features, offsets, values, d._block_sizes, d._dist_codes, d._is_sequence, side)

    # ------------------------------------------------------------------------------ the three stages of a forward
    # (what TorchRec's ShardedModule calls input_dist / compute / output_dist; forward() below runs them back to back)
    def dist_input(self, values: torch.Tensor, offsets: torch.Tensor, collapse_batch: bool = False,
                   lengths: Optional[torch.Tensor] = None):
        # (the dist only needs the offsets; lengths are derived on the device where a caller has none)
        nx = None if collapse_batch or offsets is None else self._native_for(values, offsets)
        if nx is not None:
            return nx.input_finish(self._native_begin(nx, values, offsets, None))
        return self.input_dist(lengths, values, collapse_batch, offsets=offsets)

    def dist_input_async(self, values: torch.Tensor, offsets: torch.Tensor, collapse_batch: bool = False,
                         lengths: Optional[torch.Tensor] = None, two_phase: bool = False) -> PendingKeys:
        """The input dist of a LATER batch on the exchange stream (a side HIP stream), so that bucketize, the two
        all-to-alls and the one host read of their sizes run under whatever the caller has queued on its own stream -- the
        local lookup / backward of the current batch.  (north star: "all-to-all ... overlapped with local lookup on a side
        HIP stream"; reference precedent: the data-dist stream of train_pipeline.py:155-170,589-597.)  On the CPU (gloo
        tests) there are no streams: the dist runs in place, which still exercises the reordered schedule."""
        two_phase = two_phase and not self.fixed_capacity and not collapse_batch
        if not values.is_cuda:
            if two_phase:
                return PendingKeys(None, dist=self.input_dist,
                                   state=self.input_dist(lengths, values, collapse_batch, offsets=offsets, two_phase=True))
            return PendingKeys(self.dist_input(values, offsets, collapse_batch, lengths))
        if self._comm is None:
            # HIGH priority: the exchange stream's kernels are small (bucketize, two scans, the RCCL copies: ~35 us alone) and
            # the host read of the step hangs on them -- queued behind the lookup's full-chip kernels they took 140 us and
            # arrived after the backward (profiles/r05_sharded_w1_timeline_before.txt)
            self._comm = torch.cuda.Stream(device=values.device, priority=-1)
        nx = None if collapse_batch or offsets is None else self._native_for(values, offsets)
        if nx is not None:
            # one C call: the exchange stream is ordered behind the caller's inside it.  The second half (the host read of the
            # key counts, the key exchange) follows in finish() -- at once, or (two_phase) when the caller has queued its work
            for t in (values, offsets):
                t.record_stream(self._comm)
            pend = PendingKeys(None, state=self._native_begin(nx, values, offsets, self._comm), native=nx)
            if not two_phase:
                pend.finish()
            return pend
        cur = current_torch_stream()
        self._comm.wait_stream(cur)       # the batch tensors were produced on the caller's stream
        for t in (values, offsets, lengths):
            if t is not None and t.is_cuda:
                t.record_stream(self._comm)   # ... and are read on the exchange stream: the allocator must not recycle them earlier
        if two_phase:
            with on_stream(self._comm):
                state = self.input_dist(lengths, values, collapse_batch, offsets=offsets, two_phase=True)
            return PendingKeys(None, dist=self.input_dist, state=state, comm=self._comm, consumer=cur)
        with on_stream(self._comm):
            sk = self.dist_input(values, offsets, collapse_batch, lengths)
            ev = torch.cuda.Event()
            ev.record(self._comm)
        for t in (sk.lengths, sk.offsets, sk.values, sk.recv_offsets, sk.unbucketize_permute):
            if t is not None:
                t.record_stream(cur)      # allocated on the exchange stream, consumed on the caller's
        return PendingKeys(sk, ev)

    def lookup(self, sk, train: bool = True):
        """-> (local output: pooled partial sums [W*B, total_D] fp32 or rows [n_recv, D]; local context)"""
        return self.local.forward(sk.values, sk.offsets, train)

    def dist_output(self, sk, out_local: torch.Tensor, bucketized: bool = False) -> torch.Tensor:
        W, B = self.world, sk.batch_size
        if self.pooled:
            # out_local [W*B, total_D] partial sums (fp32, or already in the wire type): block p belongs to rank p's samples
            assert out_local.size(0) == W * B
            wire = self.wire_dtype or torch.float32
            send = out_local.contiguous() if out_local.dtype == wire else out_local.to(wire)
            if self._nx is not None and send.is_cuda:
                return self._nx.output_pooled(send, self.out_dtype)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.pg)
            # (the sum reads the chunks in the wire type and accumulates in fp32: no conversion pass on either side)
            return self.ops.sum_chunks(recv.view(W, B * out_local.size(1)), self.out_dtype).view(B, out_local.size(1))
        D = out_local.size(1)
        # rows come out in (f, src, b) order; send them back in the order they arrived: (src, f, b)
        rows_sfb = self.ops.permute_bags(sk.num_features, W, B, sk.offsets, sk.recv_offsets, out_local.contiguous())
        if self._nx is not None and rows_sfb.is_cuda:
            back = self._nx.alltoallv_rows(rows_sfb, sk.recv_splits, sk.send_splits)
        else:
            back = torch.empty(sum(sk.send_splits), D, dtype=out_local.dtype, device=out_local.device)
            dist.all_to_all_single(back, rows_sfb, [s for s in sk.send_splits], [r for r in sk.recv_splits], group=self.pg)
        out = back if bucketized else self.ops.gather_rows(back, sk.unbucketize_permute)
        return out if out.dtype == self.out_dtype else out.to(self.out_dtype)

    def dist_grads(self, sk, grads: torch.Tensor, bucketized: bool = False) -> torch.Tensor:
        """gradients of the stage-3 output -> gradients of the local lookup's output (the reverse exchange)"""
        W, B = self.world, sk.batch_size
        grads = grads.contiguous()
        if self.pooled:
            if self._nx is not None and grads.is_cuda:
                return self._nx.allgather(grads)
            g_all = torch.empty(W * grads.size(0), grads.size(1), dtype=grads.dtype, device=grads.device)
            dist.all_gather_into_tensor(g_all, grads, group=self.pg)
            return g_all
        if bucketized:
            g_send = grads
        elif self.fixed_capacity:
            # padded layout: the gradient rows go to the positions of their keys, the padding rows stay zero
            g_send = torch.zeros(sum(sk.send_splits), grads.size(1), dtype=grads.dtype, device=grads.device)
            g_send.index_copy_(0, sk.unbucketize_permute, grads)
        else:
            perm = sk.unbucketize_permute
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel(), dtype=perm.dtype, device=perm.device)
            g_send = self.ops.gather_rows(grads, inv)  # bucketized order
        if self._nx is not None and g_send.is_cuda:
            g_recv = self._nx.alltoallv_rows(g_send, sk.send_splits, sk.recv_splits)
        else:
            g_recv = torch.empty(sum(sk.recv_splits), grads.size(1), dtype=grads.dtype, device=grads.device)
            dist.all_to_all_single(g_recv, g_send, [r for r in sk.recv_splits], [s for s in sk.send_splits], group=self.pg)
        return self.ops.permute_bags(W, sk.num_features, B, sk.recv_offsets, sk.offsets, g_recv)

    # ------------------------------------------------------------------------------ forward
    def forward(self, values: torch.Tensor, offsets: torch.Tensor, train: bool = True, collapse_batch: bool = False,
                lengths: Optional[torch.Tensor] = None, bucketized: bool = False):
        """bucketized (sequence mode): return the rows in EXCHANGE order (as they come back from the owners) instead of
        gathering them into key order -- the caller composes `ctx.keys.unbucketize_permute` into its own index."""
        sk = self.dist_input(values, offsets, collapse_batch, lengths)
        out_local, lctx = self.lookup(sk, train)
        out = self.dist_output(sk, out_local, bucketized)
        return out, _ShardedCtx(sk, lctx, sum(sk.send_splits))

    # ------------------------------------------------------------------------------ backward
    def backward(self, ctx: _ShardedCtx, grads: torch.Tensor, bucketized: bool = False) -> None:
        """bucketized (sequence mode): `grads` rows are already in exchange order."""
        self.local.backward(ctx.local_ctx, self.dist_grads(ctx.keys, grads, bucketized))


class OverlappedSteps:
    """Training steps of a row-wise sharded lookup with the input dist of batch i+1 in flight while batch i computes:

        forward(batch i, next_batch = batch i+1):
            keys of batch i (already exchanged) -> local lookup queued on the caller's stream
            input dist of batch i+1 issued on the exchange stream          <- runs under the lookup / backward of batch i
            output dist of batch i
        backward(batch i) ...

    The key exchange touches no table state, so moving it ahead of the previous batch's backward changes nothing in the
    results (tested against the plain schedule over gloo, tests/test_sharded_cpu.py)."""

    def __init__(self, lookup: RowWiseShardedLookup):
        self.lookup = lookup
        self._pending: Optional[PendingKeys] = None

    def prefetch(self, values: torch.Tensor, offsets: torch.Tensor) -> None:
        self._pending = self.lookup.dist_input_async(values, offsets)

    def forward(self, values: torch.Tensor, offsets: torch.Tensor, train: bool = True, next_batch=None):
        lk = self.lookup
        pend = self._pending if self._pending is not None else lk.dist_input_async(values, offsets)
        self._pending = None
        sk = pend.wait()
        if next_batch is not None:
            # first half only: bucketize, lengths exchange, key counts on their way to pinned memory.  The host read and the
            # key exchange follow in backward(), when this step's work is queued -- the host never sits waiting for the read.
            # Issued BEFORE this batch's lookup is queued: the exchange stream is ordered behind what the caller's stream
            # holds at this point (the next batch was produced there), and that must not include the lookup it is to run under
            self._pending = lk.dist_input_async(*next_batch, two_phase=True)
        out_local, lctx = lk.lookup(sk, train)
        out = lk.dist_output(sk, out_local)
        return out, _ShardedCtx(sk, lctx, sum(sk.send_splits))

    def backward(self, ctx: _ShardedCtx, grads: torch.Tensor) -> None:
        # second half of the next batch's input dist FIRST: its key counts were launched before this batch's lookup and are
        # long written; the key exchange then runs on the exchange stream UNDER this backward.  Issued behind it, the small
        # RCCL kernel waited for a slot among the backward's blocks and the next lookup waited for the keys
        # (profiles/r05_sharded_w1_timeline_before.txt: 65 us for an 8 us copy, 26 us of idle GPU behind it).
        # (An exchange THREAD for the input dist was tried as well -- the hand-over between two Python threads cost more
        #  than the ~80 us of host work it took off this one: 0.21-0.23 ms against 0.19, not kept.)
        # ... but only when that does not make the host WAIT for the counts (the exchange stream may be behind): asked again
        # behind the backward's launches, and at the latest the next forward finishes it (PendingKeys.wait) -- a host that
        # blocked here cost more than the late key exchange does (0.190 vs 0.179 ms per forced-W=1 step)
        # (the c10d sequence -- CPU tensors, MI355_NATIVE_EXCHANGE=0 -- has no such query: its host read stays behind the backward)
        # W > 1: the issue point must be the SAME on every rank.  The key exchange runs on the input communicator, the
        # backward's all-gather on the output communicator; ranks that enqueue the two in different relative orders (an
        # event query answers differently per rank) can deadlock when both streams drain through one hardware queue.  So
        # with peers the second half always goes out HERE, before the backward, whatever the query says (the host may wait
        # for the counts: they were launched before this batch's lookup); the opportunistic form is the one-rank path only.
        pend = self._pending
        if pend is not None and pend._native is not None and self.lookup.world > 1:
            pend.finish()
            self.lookup.backward(ctx, grads)
            return
        early = pend is not None and pend._native is not None and pend.try_finish()
        self.lookup.backward(ctx, grads)
        if pend is not None and not early:
            if pend._native is not None:
                pend.try_finish()
            else:
                pend.finish()


class RowWiseShardedPooledRows:
    """Pooled lookup that exchanges ROWS, not partial sums: dedup locally, run the sequence exchange on the
    unique keys, pool locally from the returned rows.  Per rank and step it moves Nu_local*D elements each way
    where the partial-sum dist moves W*B*total_D -- smaller whenever bags are short relative to W (C2 at W=8:
    162 K rows vs 524 K), and the pooled result is bit-identical to the single-GPU one (same fp32 rows summed in
    the same bag order).  The reference offers the same idea for sequence embeddings (`use_index_dedup`,
    shard/embedding.py:183-275); for pooled EBC it always reduce-scatters.

    The unique keys of table t are cut into pseudo-bags of `chunk` keys so that the bag-oriented bucketize /
    recat machinery (wave per bag) sees many small bags instead of T huge ones."""

    def __init__(self, local_seq, feature_table_map: List[int], table_hash_sizes: List[int], dims: List[int],
                 combiner: int = 0, pg=None, device=None, out_dtype=torch.float32,
                 dist_type_per_table: Optional[Sequence[str]] = None, ops=None, chunk: int = 64):
        T = len(table_hash_sizes)
        assert all(d == dims[0] for d in dims), "rows-back pooled mode needs one embedding dim (use the partial-sum dist)"
        self.dim = dims[0]
        self.F = len(feature_table_map)
        self.T = T
        self.combiner = combiner
        self.chunk = chunk
        self.out_dtype = out_dtype
        self.ops = ops or HipOps()
        tof, old = [], -1
        for i, t in enumerate(feature_table_map):
            assert t >= old, "features must be grouped by table"
            if t != old:
                tof.append(i)
                old = t
        tof.append(self.F)
        self.feature_offsets = torch.tensor(tof, dtype=torch.int64, device=device)
        self.inner = RowWiseShardedLookup(local_seq, T, table_hash_sizes, pooled=False, pg=pg, device=device,
                                          out_dtype=torch.float32, dist_type_per_feature=dist_type_per_table,
                                          ops=self.ops)
        self.world, self.rank = self.inner.world, self.inner.rank

    def forward(self, values: torch.Tensor, offsets: torch.Tensor, train: bool = True):
        B = (offsets.numel() - 1) // self.F
        ukeys, rev, uoff, aux = self.ops.unique(values, offsets, self.feature_offsets)
        nchunk = max(1, (values.numel() + self.chunk - 1) // self.chunk)
        lengths, u_offsets = self.ops.chunk_bags(uoff, self.T, self.chunk, nchunk)
        # rows of the unique keys, in EXCHANGE order: the pooling index is composed with the exchange permutation
        # instead of gathering [Nu, D] rows back into unique order (and the gradients out of it in the backward)
        rows, ictx = self.inner.forward(ukeys, u_offsets, train, collapse_batch=True, lengths=lengths, bucketized=True)
        perm = ictx.keys.unbucketize_permute
        idx = self.ops.compose(perm, rev)
        out = self.ops.pool(rows, idx, offsets, B, self.combiner, self.F * self.dim, None, self.dim, self.out_dtype)
        return out, (ictx, idx, offsets, B, rows.size(0), aux, perm)

    def backward(self, ctx, grads: torch.Tensor) -> None:
        ictx, idx, offsets, B, nu, aux, perm = ctx
        if aux is not None:
            cnt, rank, uoff = aux
            aux = (self.ops.permute_counts(cnt, perm, nu), rank, uoff)
        ug = self.ops.reduce_grads(idx, grads.contiguous(), nu, B, self.dim, offsets, None, self.combiner, aux)
        self.inner.backward(ictx, ug, bucketized=True)


class _ModuleLocal:
    """Adapter: BatchedDynamicEmbeddingTablesV2 as the `local` of RowWiseShardedLookup."""

    def __init__(self, module):
        from .dynamicemb_config import DynamicEmbPoolingMode

        # partial pooled sums of the shards are ADDED by the output dist: a local MEAN would divide by the local bag
        # length on every shard.  Pool with SUM here and divide by the global bag length after the exchange
        # (dynamicemb/shard/embeddingbag.py does), as TorchRec applies the mean after its reduce-scatter.
        if module.pooling_mode == DynamicEmbPoolingMode.MEAN:
            raise ValueError("the local module of a row-wise sharded pooled lookup must pool with SUM (apply the mean "
                             "after the output dist)")
        self.module = module

    def forward(self, values, offsets, train):
        return self.module._forward_impl(values, offsets, train=train)

    def backward(self, ctx, grads):
        self.module._backward_impl(ctx, grads)


def shard_capacity(rows: int, world: int, bucket_capacity: int = 128) -> int:
    """Per-rank table capacity: ceil(N / W) aligned up to the bucket capacity."""
    per = (rows + world - 1) // world
    return (per + bucket_capacity - 1) // bucket_capacity * bucket_capacity


class ShardedPooledLookup:
    """bench.py's N>1 path: one `rows` x `dim` table, row-wise sharded over the world, SUM pooling, SGD.
    Every rank holds ceil(rows/W) rows in its own HBM.  mode: "partial" = the reference's dist (partial sums
    for the global batch), "rows" = dedup + row exchange + local pooling, "auto" = whichever moves less."""

    def __init__(self, rows: int, dim: int, device, world: int, rank: int, lr: float = 0.1,
                 out_dtype=torch.bfloat16, dist_type: str = "roundrobin", mode: str = "auto",
                 keys_per_step: Optional[int] = None, batch: Optional[int] = None, wire_dtype="auto",
                 capacity_factor: Optional[float] = None):
        from .batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
        from .dynamicemb_config import (DynamicEmbInitializerArgs, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                        DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType)

        opt = DynamicEmbTableOptions(
            dim=dim, max_capacity=shard_capacity(rows, world), index_type=torch.int64, embedding_dtype=torch.float32,
            score_strategy=DynamicEmbScoreStrategy.TIMESTAMP,
            initializer_args=DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.UNIFORM, lower=-0.01, upper=0.01))
        if mode == "auto":
            mode = self.choose_mode(world, keys_per_step, batch, dim, out_dtype, wire_dtype)
        self.mode = mode
        if wire_dtype == "auto":
            wire_dtype = torch.bfloat16 if out_dtype == torch.bfloat16 else torch.float32
        # partial sums leave the pooling kernel in the wire type (fp32 accumulation inside it, one rounding at its store):
        # no conversion pass between the lookup and the exchange
        local_dtype = wire_dtype if (mode == "partial" and wire_dtype is not None) else torch.float32
        module = BatchedDynamicEmbeddingTablesV2(
            [opt], pooling_mode=DynamicEmbPoolingMode.SUM if mode == "partial" else DynamicEmbPoolingMode.NONE,
            output_dtype=local_dtype, device=device, optimizer=EmbOptimType.SGD, learning_rate=lr)
        module.train()
        # with RCCL's streams in the process the side stream of the early CSR build shares a hardware queue with the main
        # one and serialises (measured: no gain, +10 us of event edges): group in the backward here
        module._early_csr = False
        self.module = module
        self._steps = None
        if mode == "partial":
            self.impl = RowWiseShardedLookup(_ModuleLocal(module), 1, [rows], pooled=True, device=device,
                                             out_dtype=out_dtype, dist_type_per_feature=[dist_type], wire_dtype=wire_dtype,
                                             capacity_factor=capacity_factor, expected_keys=keys_per_step)
            self._steps = OverlappedSteps(self.impl)
        else:
            self.impl = RowWiseShardedPooledRows(_ModuleLocal(module), [0], [rows], [dim], combiner=0, device=device,
                                                 out_dtype=out_dtype, dist_type_per_table=[dist_type])
        assert self.impl.world == world and self.impl.rank == rank

    @staticmethod
    def choose_mode(world, keys_per_step, batch, dim, out_dtype=torch.bfloat16, wire_dtype="auto") -> str:
        """xGMI is a full mesh: what bounds an exchange is the bytes on ONE peer link per step, not the total.
        partial: the [B, D] block of partial sums out (fp32, or bf16 on the default wire of a bf16 output) + the [B, D]
        gradient block of the all-gather back, per peer, whatever W is.  rows: the unique rows and their fp32 gradients of the keys a peer owns, ~ 0.45 Nt / W rows each
        way -- it shrinks with W but its two-level dedup / reduce costs ~0.23 ms more compute per step (measured at
        W = 1 on one MI355X), priced here at an effective 100 GB/s per link.  C2: fp32 wire -> partial up to W = 4, rows from
        W = 8; bf16 wire (the default with bf16 outputs) -> partial at every W of one node."""
        if not (keys_per_step and batch):
            return "partial"
        o = torch.empty((), dtype=out_dtype).element_size()
        if wire_dtype == "auto":
            wire_dtype = torch.bfloat16 if out_dtype == torch.bfloat16 else torch.float32
        w = 4 if wire_dtype is None else torch.empty((), dtype=wire_dtype).element_size()
        partial_link = batch * dim * (w + o)
        rows_link = 2 * 0.45 * keys_per_step * dim * 4 / world
        handicap = 0.23e-3 * 100e9
        return "rows" if rows_link + handicap < partial_link else "partial"

    def forward(self, values, offsets, train: bool = True, next_batch=None):
        """next_batch = (values, offsets) of the following step: its key exchange starts under this step's lookup"""
        if self._steps is not None:
            return self._steps.forward(values, offsets, train, next_batch)
        return self.impl.forward(values, offsets, train)

    def backward(self, ctx, grads):
        if self._steps is not None:
            return self._steps.backward(ctx, grads)
        self.impl.backward(ctx, grads)
