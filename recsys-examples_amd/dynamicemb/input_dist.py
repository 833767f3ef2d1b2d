"""Row-wise input dist: key -> owning rank bucketize, then the all-to-all(v) of lengths and keys.

Mirror of the reference's `dynamicemb/input_dist.py` (bucketize_kjt_before_all2all :80-173,
RwSparseFeaturesDist :199-285) plus the piece of TorchRec it calls (KJTAllToAll: lengths all-to-all,
values all-to-all-v, recat to feature-major -- third party, not under /root/reference; its contract is
restated from the call site and from SURVEY.md 8(e)).

There is no TorchRec here, so a jagged batch is the plain pair (lengths [F*B] feature-major, values).
The collectives are `torch.distributed` (RCCL on the GPU, gloo in the CPU tests).  All compute around
them (bucketize, bag permutation) goes through an `ops` backend: `HipOps` (the C ABI) in production;
the CPU tests inject a numpy backend built on oracle/ -- this module itself never touches oracle/.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

DIST_TYPES = {"continuous": 0, "roundrobin": 1, "hash_roundrobin": 2}



def _cur_stream():
    from mi355_native import current_torch_stream     # (GPU tensors only)
    return current_torch_stream()

class TorchGlue:
    """Index bookkeeping of the exchange in plain torch (device agnostic): the base of every `ops` backend.  HipOps
    overrides each method with ONE launch; the CPU test backend inherits these."""

    def exclusive_offsets(self, lengths):
        off = torch.zeros(lengths.numel() + 1, dtype=torch.int64, device=lengths.device)
        torch.cumsum(lengths, 0, out=off[1:])
        return off

    def peer_splits(self, send_offsets, recv_offsets, per_peer, world):
        """keys sent to / received from every peer (python lists): the one host read of the exchange"""
        idx = torch.arange(world + 1, device=send_offsets.device) * per_peer
        s, r = send_offsets[idx], recv_offsets[idx]
        both = torch.stack([s[1:] - s[:-1], r[1:] - r[:-1]]).cpu()
        return both[0].tolist(), both[1].tolist()

    def peer_splits_begin(self, send_offsets, recv_offsets, per_peer, world):
        return self.peer_splits(send_offsets, recv_offsets, per_peer, world)

    def peer_splits_end(self, handle):
        return handle

    def chunk_bags(self, unique_offsets, num_tables, chunk, num_chunks):
        """per-table unique-key lists cut into `num_chunks` pseudo-bags of <= chunk keys -> (lengths, offsets)"""
        lo = unique_offsets[:-1].view(num_tables, 1)
        cnt = (unique_offsets[1:] - unique_offsets[:-1]).view(num_tables, 1)
        steps = torch.arange(num_chunks + 1, dtype=torch.int64, device=unique_offsets.device).view(1, -1) * chunk
        cut = torch.minimum(steps, cnt)                        # [T, num_chunks + 1]
        lengths = (cut[:, 1:] - cut[:, :-1]).reshape(-1)
        offsets = torch.cat([(lo + cut[:, :-1]).reshape(-1), unique_offsets[-1:]])
        return lengths, offsets

    def pad_for_exchange(self, new_values, new_offsets, new_lengths, perm, world, per_peer, cap, pad_key=-1):
        """Fixed-capacity layout of a bucketized key stream: peer p's slot [p*cap, (p+1)*cap) holds its keys plus padding
        with an INVALID key (the table kernels give it no slot: zero row, no update).  The padding is spread EVENLY over
        the peer's bags, behind each bag's real keys -- one giant padding bag would be walked by a single lane group of the
        pooled gather (measured: 20 ms).  -> (values [W*cap], lengths [W*per_peer], positions of the original keys in the
        padded layout | None, overflow flag [1] bool: some peer got more than cap)"""
        dev = new_values.device
        n = new_values.numel()
        peer_off = new_offsets[torch.arange(world + 1, device=dev) * per_peer]
        cnt = peer_off[1:] - peer_off[:-1]
        overflow = (cnt > cap).any().view(1)
        pad = (cap - cnt).clamp_(min=0)
        bag_pad = (pad // per_peer).view(world, 1) + (torch.arange(per_peer, device=dev).view(1, -1) < (pad % per_peer).view(world, 1))
        lens2 = new_lengths.view(world, per_peer) + bag_pad
        off2 = torch.cumsum(lens2, 1) - lens2 + (torch.arange(world, device=dev) * cap).view(world, 1)   # start of every bag
        off2 = off2.reshape(-1)
        vals = torch.full((world * cap,), pad_key, dtype=new_values.dtype, device=dev)
        dst = None
        if n:
            bag = torch.repeat_interleave(torch.arange(world * per_peer, device=dev), new_lengths, output_size=n)
            dst = off2[bag] + (torch.arange(n, device=dev) - new_offsets[bag])
            vals[dst.clamp(max=world * cap - 1)] = new_values       # (an overflowing peer spills into garbage: flagged)
        perm_p = None
        if perm is not None:
            perm_p = dst[perm] if n else perm
        return vals, lens2.reshape(-1), perm_p, overflow

    def compose(self, perm, index):
        """perm[index]: lets a consumer read rows in exchange order instead of gathering them back first"""
        return perm[index]

    def permute_counts(self, counts, perm, n):
        """out[perm[u]] = counts[u], u < n (per-unique occurrence counts re-keyed to exchange order)"""
        if counts is None:
            return None
        out = torch.zeros(n, dtype=counts.dtype, device=counts.device)
        out[perm] = counts[:n]
        return out


class HipOps(TorchGlue):
    """Compute backend of the sharded path: every method is one or two C-ABI launches."""

    def bucketize(self, offsets, values, block_sizes, world, sequence, dist_types):
        """-> (new_lengths [W*F*B], new_offsets [W*F*B+1], new_values, unbucketize_permute | None)"""
        from mi355_native import check, lib, ptr, stream

        FB = offsets.numel() - 1
        B = FB // block_sizes.numel()
        dev = values.device
        new_lengths = torch.empty(world * FB, dtype=torch.int64, device=dev)
        new_offsets = torch.empty(world * FB + 1, dtype=torch.int64, device=dev)
        new_values = torch.empty_like(values)
        perm = torch.empty(values.numel(), dtype=torch.int64, device=dev) if sequence else None
        check(lib().mi355_block_bucketize(world, FB, B, ptr(offsets), ptr(values), ptr(block_sizes), ptr(dist_types), None,
                                          ptr(new_lengths), ptr(new_offsets), ptr(new_values), None, ptr(perm), stream()),
              "block_bucketize")
        return new_lengths, new_offsets, new_values, perm

    def exclusive_offsets(self, lengths):
        from mi355_native import check, lib, ptr, stream

        off = torch.empty(lengths.numel() + 1, dtype=torch.int64, device=lengths.device)
        check(lib().mi355_exclusive_offsets(ptr(lengths), lengths.numel(), ptr(off), stream()), "exclusive_offsets")
        return off

    def peer_splits_begin(self, send_offsets, recv_offsets, per_peer, world):
        """launch the per-peer key counts into pinned host memory (a small ring: a later dist may start before an earlier one
        has been read); peer_splits_end() waits for exactly this launch and reads them"""
        from mi355_native import check, current_torch_stream, lib, ptr, stream

        ring = getattr(self, "_splits_ring", None)
        if ring is None or ring[0].numel() < 2 * world:
            ring = self._splits_ring = [torch.empty(2 * world, dtype=torch.int64).pin_memory() for _ in range(4)]
            self._splits_next = 0
        buf = ring[self._splits_next % len(ring)]
        self._splits_next += 1
        # the kernel writes straight into pinned host memory; the host waits for an event, not for a copy
        check(lib().mi355_peer_splits(ptr(send_offsets), ptr(recv_offsets), per_peer, world, ptr(buf), stream()), "peer_splits")
        ev = torch.cuda.Event()
        ev.record(current_torch_stream())
        return buf, ev, world

    def peer_splits_end(self, handle):
        buf, ev, world = handle
        ev.synchronize()
        v = buf[:2 * world].tolist()
        return v[:world], v[world:]

    def peer_splits(self, send_offsets, recv_offsets, per_peer, world):
        return self.peer_splits_end(self.peer_splits_begin(send_offsets, recv_offsets, per_peer, world))

    def chunk_bags(self, unique_offsets, num_tables, chunk, num_chunks):
        from mi355_native import check, lib, ptr, stream

        n = num_tables * num_chunks
        lengths = torch.empty(n, dtype=torch.int64, device=unique_offsets.device)
        offsets = torch.empty(n + 1, dtype=torch.int64, device=unique_offsets.device)
        check(lib().mi355_chunk_bags(ptr(unique_offsets), num_tables, chunk, num_chunks, ptr(lengths), ptr(offsets), stream()),
              "chunk_bags")
        return lengths, offsets

    def permute_lengths(self, S, F, B, lengths):
        from mi355_native import check, lib, ptr, stream

        if S == 1 or F == 1:
            return lengths  # (s, f, b) and (f, s, b) are the same order
        out = torch.empty_like(lengths)
        check(lib().mi355_permute_lengths(S, F, B, ptr(lengths), ptr(out), stream()), "permute_lengths")
        return out

    def permute_bags(self, S, F, B, in_offsets, out_offsets, data):
        from mi355_native import check, lib, ptr, stream

        if S == 1 or F == 1:
            return data
        out = torch.empty_like(data)
        eb = data.element_size() * (data.size(1) if data.dim() == 2 else 1)
        check(lib().mi355_permute_bags(S, F, B, eb, data.size(0), ptr(in_offsets), ptr(out_offsets), ptr(data), ptr(out),
                                       stream()), "permute_bags")
        return out

    def sum_chunks(self, x, out_dtype):
        from mi355_native import check, dt, lib, ptr, stream

        out = torch.empty(x.shape[1:], dtype=out_dtype, device=x.device)
        if x.dtype == torch.float32:
            check(lib().mi355_sum_chunks(ptr(x), x.size(0), out.numel(), ptr(out), dt(out_dtype), stream()), "sum_chunks")
        else:     # chunks in the wire type: read as they arrived, fp32 accumulation
            check(lib().mi355_sum_chunks_typed(ptr(x), dt(x.dtype), x.size(0), out.numel(), ptr(out), dt(out_dtype), stream()),
                  "sum_chunks")
        return out

    def gather_rows(self, src, index):
        import dynamicemb_extensions as ext

        out = torch.empty(index.numel(), src.size(1), dtype=src.dtype, device=src.device)
        if index.numel():
            ext.gather_embedding(src, out, index)
        return out

    # ---- local dedup / pooling around the exchange (rows-back pooled mode, sharded.py) ----
    def unique(self, keys, offsets, feature_offsets):
        """-> (unique_keys [Nt] (first Nu valid), reverse [Nt], unique_offsets [T+1], aux) per-table dedup; `aux` carries
        the per-unique counts / per-key ranks from which reduce_grads builds its CSR without a histogram pass."""
        import dynamicemb_extensions as ext

        T = feature_offsets.numel() - 1
        rng = ext.get_table_range(offsets, feature_offsets)
        ukeys, rev, uoff, cnt, rank = ext.segmented_unique_csr(keys, rng, T)
        return ukeys, rev, uoff, (cnt, rank, uoff)

    def pool(self, rows, reverse, offsets, batch_size, combiner, total_D, D_offsets, max_D, out_dtype):
        import dynamicemb_extensions as ext

        out = torch.empty(batch_size, total_D, dtype=out_dtype, device=rows.device)
        ext.gather_embedding_pooled(rows, out, reverse, offsets, combiner, total_D, batch_size, D_offsets=D_offsets,
                                    max_D=max_D)
        return out

    def reduce_grads(self, reverse, grads, num_unique, batch_size, dim, offsets, D_offsets, combiner, aux=None):
        import dynamicemb_extensions as ext

        if aux is None:
            return ext.reduce_grads(reverse, grads, num_unique, batch_size, dim, offsets=offsets, D_offsets=D_offsets,
                                    combiner=combiner, out_dtype=torch.float32)
        cnt, rank, uoff = aux
        n = reverse.numel()
        out = torch.empty(num_unique, dim, dtype=torch.float32, device=grads.device)
        if n == 0 or num_unique == 0:
            return out
        ptr_t, csr, hot = ext.group_by_unique_csr(cnt, rank, reverse, num_unique, offsets, nu_dev=uoff[-1:], dim=dim)
        ext.backward_fused(ptr_t, csr, n, num_unique, grads.contiguous(), batch_size, dim, combiner, offsets, D_offsets,
                           weight_dtype=torch.float32, opt_kind=0, out=out, round_grad=False, nu_dev=uoff[-1:], hot=hot)
        return out


def exclusive_offsets(lengths: torch.Tensor) -> torch.Tensor:
    off = torch.zeros(lengths.numel() + 1, dtype=torch.int64, device=lengths.device)
    torch.cumsum(lengths, 0, out=off[1:])
    return off


@dataclass
class ShardedKeys:
    """What a rank holds after the input dist: the keys it owns, for the GLOBAL batch."""
    lengths: torch.Tensor          # [F * W * B] feature-major, batch index = src_rank * B + b
    offsets: torch.Tensor          # [F * W * B + 1]
    values: torch.Tensor           # [n_recv]
    recv_offsets: torch.Tensor     # [W * F * B + 1] offsets of the stream as received, (src, f, b) order
    send_splits: List[int]         # keys sent to each rank
    recv_splits: List[int]         # keys received from each rank
    unbucketize_permute: Optional[torch.Tensor]  # [n_local] (sequence mode)
    batch_size: int                # local B
    num_features: int


def bucketize_before_all2all(lengths, values, num_buckets, block_sizes, output_permute=False,
                             dist_type_per_feature: Optional[Sequence[str]] = None, ops=None, offsets=None):
    """bucketize_kjt_before_all2all (input_dist.py:80-173): -> (new_lengths [W*F*B], new_values, permute,
    new_offsets [W*F*B+1])."""
    ops = ops or HipOps()
    F = block_sizes.numel()
    if dist_type_per_feature is None:
        dist_type_per_feature = ["continuous"] * F
    codes = []
    for d in dist_type_per_feature:
        if d not in DIST_TYPES:
            raise ValueError("Not support dist type of ", d)
        codes.append(DIST_TYPES[d])
    dist_t = torch.tensor(codes, dtype=torch.int32, device=values.device)
    if offsets is None:
        offsets = ops.exclusive_offsets(lengths.view(-1).to(torch.int64))
    nl, no, nv, perm = ops.bucketize(offsets, values, block_sizes.to(values.device), num_buckets, output_permute, dist_t)
    return nl, nv, perm, no


class RwSparseFeaturesDist:
    """RwSparseFeaturesDist (input_dist.py:199-285) + KJTAllToAll, for equal local batch sizes."""

    def __init__(self, pg, num_features: int, feature_hash_sizes: List[int], device=None, is_sequence: bool = False,
                 dist_type_per_feature: Optional[Sequence[str]] = None, ops=None, capacity_factor: Optional[float] = None,
                 expected_keys: Optional[int] = None):
        """capacity_factor: None = exact all-to-all-v (one host read of the per-peer key counts per step, as KJTAllToAll).
        A number = FIXED-CAPACITY exchange: every peer slot carries ceil(factor * expected_keys / W) keys (`expected_keys`: the
        keys of one rank's batch, the SAME number on every rank -- the slot size must agree), padded with invalid keys, so
        every size is known on the host and the step has no device-to-host read at all (and a fixed launch sequence).  A
        peer whose share exceeds the slot sets a sticky overflow flag that `check_overflow()` turns into an error; with
        hash_roundrobin routing factor 2 covers a Zipf-0.99 stream (the hottest key drags 6 % of a batch to one rank)."""
        self._cap_factor = capacity_factor
        if capacity_factor is not None and not expected_keys:
            raise ValueError("capacity_factor needs expected_keys (keys per rank and step, identical on every rank)")
        self._expected_keys = expected_keys
        self._overflow = None
        self._overflow_host = None
        self._overflow_event = None
        self._pg = pg
        self._world_size = dist.get_world_size(pg)
        self._num_features = num_features
        self._block_sizes = torch.tensor([(h + self._world_size - 1) // self._world_size for h in feature_hash_sizes],
                                         dtype=torch.int64, device=device)
        self._is_sequence = is_sequence
        self._dist_type_per_feature = list(dist_type_per_feature) if dist_type_per_feature is not None \
            else ["roundrobin"] * num_features
        for d_ in self._dist_type_per_feature:
            if d_ not in DIST_TYPES:
                raise ValueError("Not support dist type of ", d_)
        self._ops = ops or HipOps()
        self._dist_codes = None
        self.unbucketize_permute_tensor = None

    def forward(self, lengths: torch.Tensor, values: torch.Tensor, collapse_batch: bool = False,
                offsets: Optional[torch.Tensor] = None, two_phase: bool = False):
        """two_phase: return after the launch of the per-peer key counts (everything up to the one host read of the step);
        `finish(state)` completes the exchange.  A caller that has other work to queue in between never waits for the
        read.  (Only the exact exchange has a second phase; the fixed-capacity one returns the keys directly.)"""
        """collapse_batch: the batch dimension is only a local chunking of per-feature key lists (it may differ
        between ranks); the exchange then carries one bag per (rank, feature).  `offsets` (optional): the exclusive
        offsets of `lengths` when the caller already has them."""
        W, F, ops = self._world_size, self._num_features, self._ops
        if lengths is None:
            assert offsets is not None, "lengths or offsets required"
            nbags = offsets.numel() - 1
        else:
            lengths = lengths.view(-1)
            if lengths.dtype != torch.int64:
                lengths = lengths.to(torch.int64)
            nbags = lengths.numel()
        assert nbags % F == 0
        B = nbags // F
        if self._dist_codes is None or self._dist_codes.device != values.device:
            self._dist_codes = torch.tensor([DIST_TYPES[d] for d in self._dist_type_per_feature], dtype=torch.int32,
                                            device=values.device)
            self._block_sizes = self._block_sizes.to(values.device)
        if offsets is None:
            offsets = ops.exclusive_offsets(lengths)
        new_lengths, new_offsets, new_values, perm = ops.bucketize(offsets, values, self._block_sizes, W, self._is_sequence,
                                                                   self._dist_codes)
        self.unbucketize_permute_tensor = perm
        if self._cap_factor is not None and not collapse_batch:
            return self._forward_fixed(new_lengths, new_offsets, new_values, perm, values, B)
        if collapse_batch:
            # one bag per (peer, feature): its offsets are the bucketized offsets at the (peer, feature) boundaries
            idx = torch.arange(W * F + 1, device=values.device) * B
            new_offsets = new_offsets[idx]
            new_lengths = new_offsets[1:] - new_offsets[:-1]
            B = 1
        # lengths all-to-all: equal splits of F*B per peer
        recv_lengths = torch.empty_like(new_lengths)
        dist.all_to_all_single(recv_lengths, new_lengths, group=self._pg)
        recv_offsets = ops.exclusive_offsets(recv_lengths)
        # key counts per peer (the one host read of the step, as in KJTAllToAll): launched here, read in finish()
        handle = ops.peer_splits_begin(new_offsets, recv_offsets, F * B, W)
        state = (handle, new_values, perm, recv_lengths, recv_offsets, B)
        if two_phase:
            return state
        return self.finish(state)

    def finish(self, state) -> ShardedKeys:
        """second half of the exact exchange: read the per-peer key counts (the host waits for the launch that wrote them,
        which by now is usually long done) and send the keys"""
        W, F, ops = self._world_size, self._num_features, self._ops
        handle, new_values, perm, recv_lengths, recv_offsets, B = state
        send_splits, recv_splits = ops.peer_splits_end(handle)
        n_send = sum(send_splits)  # == values.numel() unless the caller passed a padded key buffer
        new_values = new_values[:n_send]
        if perm is not None:
            perm = perm[:n_send]
        recv_values = torch.empty(sum(recv_splits), dtype=new_values.dtype, device=new_values.device)
        dist.all_to_all_single(recv_values, new_values, recv_splits, send_splits, group=self._pg)
        # recat (src, f, b) -> (f, src, b); the two orders coincide when there is one source or one feature
        if W == 1 or F == 1:
            fm_lengths, fm_offsets, fm_values = recv_lengths, recv_offsets, recv_values
        else:
            fm_lengths = ops.permute_lengths(W, F, B, recv_lengths)
            fm_offsets = ops.exclusive_offsets(fm_lengths)
            fm_values = ops.permute_bags(W, F, B, recv_offsets, fm_offsets, recv_values)
        return ShardedKeys(fm_lengths, fm_offsets, fm_values, recv_offsets, send_splits, recv_splits, perm, B, F)

    def _forward_fixed(self, new_lengths, new_offsets, new_values, perm, values, B):
        W, F, ops = self._world_size, self._num_features, self._ops
        self.check_overflow()
        cap = max(8, -(-int(self._cap_factor * self._expected_keys) // W))
        cap = (cap + 7) // 8 * 8
        vals, lens, perm_p, ov = ops.pad_for_exchange(new_values, new_offsets, new_lengths, perm, W, F * B, cap)
        self._overflow = ov if self._overflow is None else (self._overflow | ov)
        if ov.is_cuda:   # the flag travels to pinned memory behind the step; the host looks at it one step later
            if self._overflow_host is None:
                self._overflow_host = torch.zeros(1, dtype=torch.bool).pin_memory()
            self._overflow_host.copy_(self._overflow, non_blocking=True)
            self._overflow_event = torch.cuda.Event()
            self._overflow_event.record(_cur_stream())
        self.unbucketize_permute_tensor = perm_p
        recv_lengths = torch.empty_like(lens)
        dist.all_to_all_single(recv_lengths, lens, group=self._pg)
        recv_offsets = ops.exclusive_offsets(recv_lengths)
        recv_values = torch.empty_like(vals)
        dist.all_to_all_single(recv_values, vals, group=self._pg)        # equal splits: nothing to read back
        splits = [cap] * W
        if W == 1 or F == 1:
            fm_lengths, fm_offsets, fm_values = recv_lengths, recv_offsets, recv_values
        else:
            fm_lengths = ops.permute_lengths(W, F, B, recv_lengths)
            fm_offsets = ops.exclusive_offsets(fm_lengths)
            fm_values = ops.permute_bags(W, F, B, recv_offsets, fm_offsets, recv_values)
        return ShardedKeys(fm_lengths, fm_offsets, fm_values, recv_offsets, splits, list(splits), perm_p, B, F)

    def check_overflow(self) -> None:
        """raises if a previous fixed-capacity exchange had to drop keys (its results are invalid)"""
        bad = False
        if self._overflow is not None and not self._overflow.is_cuda:
            bad = bool(self._overflow.item())
        elif self._overflow_event is not None and self._overflow_event.query():
            bad = bool(self._overflow_host.item())
        if bad:
            raise RuntimeError("fixed-capacity key exchange overflowed: a peer's share of a batch exceeded "
                               f"capacity_factor = {self._cap_factor} x (keys / world size); raise the factor or use the "
                               "exact exchange (capacity_factor=None)")

    __call__ = forward
