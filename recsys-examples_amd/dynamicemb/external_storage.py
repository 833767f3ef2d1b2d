"""Tables backed by a user-supplied key-value store (`DynamicEmbTableOptions.external_storage`).

Reference: the HOST_PS layout of BatchedDynamicEmbeddingTablesV2 (batched_dynamicemb_tables.py:698-721) and its generic
forward / backward (batched_dynamicemb_function.py:934-1040, 1190-1290): the module de-duplicates the batch, asks the store
for the rows of the unique keys, initialises and inserts the ones it does not have, pools from the dense value buffer, and in
the backward reduces the gradients per unique key, applies the optimizer to the buffer and writes the rows back.
Everything but `Storage.find` / `Storage.insert` (user code) runs in the HIP kernels of the per-op chain: segmented unique,
row initialisation, pooled / row gather, gradient reduction, the padded-buffer optimizers.

Value rows travel to and from the store in the reference's PADDED layout (`[embedding | pad to the widest embedding | optimizer
state]`, key_value_table.py / optimizer_kernel.cuh:28-39), so tables of different widths share one buffer.

`caching=True` (the reference's CACHING_PS layout, batched_dynamicemb_tables.py:694-706): an HBM table of
`local_hbm_for_values` bytes sits in front of the store.  A key lives in exactly one place: misses are fetched from the store (or
initialised) and inserted into the cache, what the cache evicts is written back to the store, keys the cache refuses are trained in
a spill buffer and written back after their backward; `flush()` writes the whole cache back.  Hits never leave the GPU.
"""
import json
import os
from copy import deepcopy
from typing import List, Optional

import torch
from torch import nn

import dynamicemb_extensions as ext

from itertools import accumulate

from mi355_native import check, dt, lib, ptr, stream

from .batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2, _StepCtx, init_dense_rows
from .dynamicemb_config import DynamicEmbPoolingMode, DynamicEmbScoreStrategy
from .optimizer import OptimizerView
from .types import CopyMode


class _ExtStep:
    """what a forward leaves for its backward: the unique keys, their table ids, the value buffer the store returned"""


def _parse_find(r):
    """the 8-tuple of the reference's storages or the 7-tuple of its abstract class (no missing_table_ids)"""
    if len(r) == 8:
        num_missing, mkeys, midx, mtids, mscores, founds, oscores, values = r
    elif len(r) == 7:
        num_missing, mkeys, midx, mscores, founds, oscores, values = r
        mtids = None
    else:
        raise RuntimeError(f"Storage.find returned {len(r)} values (7 or 8 expected)")
    if isinstance(num_missing, torch.Tensor):
        num_missing = int(num_missing.item())
    return int(num_missing), mkeys, midx, mtids, mscores, founds, oscores, values


class ExternalStorageTables(BatchedDynamicEmbeddingTablesV2):
    """BatchedDynamicEmbeddingTablesV2 whose rows live in `table_options[i].external_storage` (a `dynamicemb.types.Storage`
    subclass, constructed here with (options, OptimizerView) as the reference does).  Same forward / backward surface;
    prefetch, table growth and admission are not available."""

    def __init__(self, table_options, table_names=None, feature_table_map=None, use_index_dedup=False, prefetch_pipeline=False,
                 pooling_mode=DynamicEmbPoolingMode.SUM, output_dtype=torch.float32, device=None, enforce_hbm=False,
                 bounds_check_mode=None, optimizer=None, stochastic_rounding=True, gradient_clipping=False, max_gradient=1.0,
                 max_norm=0.0, learning_rate=0.01, eps=1.0e-8, initial_accumulator_value=0.0, momentum=0.9, weight_decay=0.0,
                 weight_decay_mode=None, eta=0.001, beta1=0.9, beta2=0.999, counter_based_regularization=None,
                 cowclip_regularization=None, storage_mode=None, *args, **kwargs):
        nn.Module.__init__(self)
        from .batched_dynamicemb_tables import _OPT_KIND, EmbOptimType, get_optimizer_state_dim

        optimizer = optimizer if optimizer is not None else EmbOptimType.SGD
        opt0 = table_options[0]
        for o in table_options:
            assert opt0 == o, "All tables must match in grouped keys."
        if opt0.admit_strategy is not None:
            raise NotImplementedError("admission with an external storage")
        self._dynamicemb_options = table_options
        self._table_names = table_names or [f"t{i}" for i in range(len(table_options))]
        self.pooling_mode, self.output_dtype, self.use_index_dedup = pooling_mode, output_dtype, use_index_dedup
        self.index_type = opt0.index_type or torch.int64
        self.embedding_dtype = opt0.embedding_dtype or torch.float32
        self.device_ = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dims: List[int] = [o.dim for o in table_options]
        if pooling_mode == DynamicEmbPoolingMode.NONE:
            assert all(d == self.dims[0] for d in self.dims), "Sequence mode requires uniform embedding dim"
        T_ = len(table_options)
        self.feature_table_map = feature_table_map if feature_table_map is not None else list(range(T_))
        assert all(any(t == m for m in self.feature_table_map) for t in range(T_)), "Each table must have at least one feature!"
        D_offsets = [0] + list(accumulate(self.dims[t] for t in self.feature_table_map))
        self.total_D = D_offsets[-1]
        self.max_D = max(self.dims)
        self.mixed_D = self.max_D > min(self.dims)
        self.D_offsets_t = torch.tensor(D_offsets, device=self.device_, dtype=torch.int32) if self.mixed_D else None
        self.feature_num = len(self.feature_table_map)
        tof, old = [], -1
        for i, t in enumerate(self.feature_table_map):
            if t != old:
                tof.append(i)
                old = t
        tof.append(self.feature_num)
        self.table_offsets_in_feature = tof
        self.feature_offsets = torch.tensor(tof, device=self.device_, dtype=torch.int64)
        self.num_tables = T_
        if optimizer.name not in _OPT_KIND:
            raise ValueError(f"Not supported optimizer type: {optimizer}")
        self._opt_kind = _OPT_KIND[optimizer.name]
        self.optimizer_type = optimizer
        self.learning_rate, self.eps, self.beta1, self.beta2 = learning_rate, eps, beta1, beta2
        self.weight_decay, self.initial_accumulator_value = weight_decay, initial_accumulator_value
        self._iter_num = 0
        self.value_dims = [d + get_optimizer_state_dim(optimizer, d, self.embedding_dtype) for d in self.dims]
        self._score_strategy = opt0.score_strategy
        self._step, self._custom_score = 0, 0
        self.initializer_args = opt0.initializer_args
        self._seed = 1234
        self.storage_mode = "external"
        self._empty_tensor = nn.Parameter(torch.empty(10, requires_grad=True, device=self.device_, dtype=self.embedding_dtype))
        storage_options = deepcopy(list(table_options))
        for so in storage_options:
            so.local_hbm_for_values = 0
        self._storage = opt0.external_storage(storage_options, OptimizerView(self))
        self._orphan_pins, self._prefetch_states = [], ()
        self.table = self.table_host = None
        self.dims_t = torch.tensor(self.dims, dtype=torch.int64, device=self.device_)
        self.max_state = max(v - d for v, d in zip(self.value_dims, self.dims))
        self.max_V = self.max_D + self.max_state          # width of a padded value row
        # ---- caching=True: an HBM-only module of the cache's size holds the hot rows (its hash table, its flat value rows, its
        #      backward); this class walks cache -> store around it
        self._cache = None
        if opt0.caching:
            C = opt0.bucket_capacity
            eb = torch.empty((), dtype=self.embedding_dtype).element_size()
            row_bytes = [v * eb for v in self.value_dims]
            total = sum(o.max_capacity * b for o, b in zip(table_options, row_bytes))
            cache_opts = deepcopy(list(table_options))
            for o, b in zip(cache_opts, row_bytes):
                share = opt0.local_hbm_for_values * (o.max_capacity * b) // max(total, 1) if opt0.local_hbm_for_values > 0 else o.max_capacity * b
                rows = max(C, (share // b) // C * C)
                o.max_capacity = min(rows, (o.max_capacity + C - 1) // C * C)
                o.init_capacity = None
                o.external_storage, o.caching, o.local_hbm_for_values = None, False, 0
            self._cache = BatchedDynamicEmbeddingTablesV2(
                cache_opts, table_names=self._table_names, feature_table_map=self.feature_table_map, pooling_mode=pooling_mode,
                output_dtype=output_dtype, device=self.device_, optimizer=optimizer, learning_rate=learning_rate, eps=eps,
                initial_accumulator_value=initial_accumulator_value, weight_decay=weight_decay, beta1=beta1, beta2=beta2,
                storage_mode="hbm")
            self._cache._seed = self._seed

    # ------------------------------------------------------------------ scores handed to Storage.insert
    def _insert_scores(self, n: int, freq: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        s = self._score_strategy
        if s == DynamicEmbScoreStrategy.TIMESTAMP:
            return torch.full((n,), ext.device_timestamp(), dtype=torch.int64, device=self.device_)
        if s == DynamicEmbScoreStrategy.STEP:
            return torch.full((n,), self._step, dtype=torch.int64, device=self.device_)
        if s == DynamicEmbScoreStrategy.CUSTOMIZED:
            return torch.full((n,), self._custom_score, dtype=torch.int64, device=self.device_)
        if s == DynamicEmbScoreStrategy.LFU and freq is not None:
            return freq
        return None

    # ------------------------------------------------------------------ forward / backward (called by _LookupFunction / forward())
    def _forward_impl(self, indices: torch.Tensor, offsets: torch.Tensor, train: bool, prefetch_only: bool = False):
        if prefetch_only:
            raise NotImplementedError("prefetch with an external storage")
        indices = indices.contiguous()
        if indices.dtype != torch.int64:
            indices = indices.to(torch.int64)
        offsets = offsets.to(torch.int64).contiguous()
        if self._cache is not None:
            return self._forward_cached(indices, offsets, train)
        n, num_bags = indices.numel(), offsets.numel() - 1
        B = num_bags // self.feature_num
        T, D, V, dev = self.num_tables, self.max_D, self.max_V, self.device_
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        lfu = self._score_strategy == DynamicEmbScoreStrategy.LFU
        seg = ext.get_table_range(offsets, self.feature_offsets)
        _, ukeys, rev, uoff, freq = ext.segmented_unique_cuda(indices, seg, T, torch.empty(0, dtype=torch.int64, device=dev) if lfu else None)
        nu = int(uoff[T].item())      # (the store is host code: its call is a synchronisation point anyway)
        ukeys = ukeys[:nu].contiguous()
        tids = ext.expand_table_ids_cuda(uoff, nu)
        freq = freq[:nu].contiguous() if lfu else None
        if nu == 0:
            values = torch.zeros(0, V if train else D, dtype=self.embedding_dtype, device=dev)
        elif train:
            num_missing, mkeys, midx, mtids, _, _, _, values = _parse_find(self._storage.find(ukeys, tids, CopyMode.VALUE, freq))
            if num_missing > 0:
                midx = midx.to(torch.int64).contiguous()
                if mtids is None:
                    mtids = tids[midx]
                # first touch: embedding columns by the initializer (keyed by the KEY, as in the HBM tier: the same key draws the
                # same row in every storage mode), state columns at their initial value -- then the store learns the row
                self._init_padded_rows(values, midx, mkeys, mtids)
                sc = self._insert_scores(nu, freq)
                self._storage.insert(mkeys, mtids, values[midx], sc[midx] if sc is not None else None)
        else:
            num_missing, _, midx, _, _, _, _, values = _parse_find(self._storage.find(ukeys, tids, CopyMode.EMBEDDING, None))
            if num_missing > 0:
                init_dense_rows(values, midx, self._dynamicemb_options[0].eval_initializer_args)
        emb = values[:, :D] if values.size(1) != D else values
        if pooled:
            out = torch.empty(B, self.total_D, dtype=self.output_dtype, device=dev)
            combiner = 0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1
            if n == 0 or nu == 0:
                out.zero_()
            else:
                ext.gather_embedding_pooled(emb, out, rev, offsets, combiner, self.total_D, B, self.D_offsets_t, D)
        else:
            out = torch.empty(n, D, dtype=self.output_dtype, device=dev)
            if n:
                ext.gather_embedding(emb, out, rev)
        if not train:
            return out, None
        st = _ExtStep()
        st.rev, st.offsets, st.num_keys, st.batch_size, st.num_bags = rev, offsets, n, B, num_bags
        st.ukeys, st.tids, st.values, st.nu, st.freq = ukeys, tids, values, nu, freq
        self._step += 1
        return out, st

    def _init_padded_rows(self, values: torch.Tensor, midx: torch.Tensor, mkeys: torch.Tensor, mtids: torch.Tensor) -> None:
        """rows `midx` of the padded buffer `values` <- first-touch rows of the keys `mkeys` (tables `mtids`): the embedding by the
        initializer, keyed by the key; the state, behind the widest embedding, at its initial value; zeros between"""
        mode, p = self._init_params()
        eb = values.element_size()
        addr = (values.data_ptr() + midx * (values.stride(0) * eb)).contiguous()
        if not self.mixed_D:
            ext.init_rows(mode, p, self._seed, float(self.initial_accumulator_value), mkeys.to(torch.int64).contiguous(), addr,
                          values.dtype, self.max_D, values.size(1))
            return
        values[midx] = 0
        ext.init_rows(mode, p, self._seed, float(self.initial_accumulator_value), mkeys.to(torch.int64).contiguous(), addr,
                      values.dtype, self.max_D, self.max_D, table_ids=mtids.to(torch.int64).contiguous(),
                      table_emb_dims=self.dims_t, table_value_dims=self.dims_t)
        if self.max_state:
            values[midx, self.max_D:] = float(self.initial_accumulator_value)

    def _backward_impl(self, st, grads: torch.Tensor):
        if self._cache is not None:
            return self._backward_cached(st, grads)
        if st is None or st.nu == 0:
            return
        D, V = self.max_D, st.values.size(1)
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        combiner = (0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1) if pooled else -1
        g = ext.reduce_grads(st.rev, grads.contiguous(), st.nu, st.batch_size, D, st.offsets if pooled else None, self.D_offsets_t,
                             combiner, self.total_D)
        vals, tid = st.values, st.tids
        dims_t = self.dims_t
        al = all(d % 4 == 0 for d in self.dims) and V % 4 == 0
        self._iter_num += 1
        if self._opt_kind == 1:
            ext.sgd_update_for_padded_buffer(g, vals, tid, dims_t, D, V, al, self.learning_rate)
        elif self._opt_kind == 2:
            ext.adam_update_for_padded_buffer(g, vals, tid, dims_t, D, V, al, self.learning_rate, self.beta1, self.beta2, self.eps,
                                              self.weight_decay, self._iter_num)
        elif self._opt_kind == 3:
            ext.adagrad_update_for_padded_buffer(g, vals, tid, dims_t, D, V, al, self.learning_rate, self.eps)
        else:
            ext.rowwise_adagrad_for_padded_buffer(g, vals, tid, dims_t, D, V, al, self.learning_rate, self.eps)
        self._storage.insert(st.ukeys, tid, vals, self._insert_scores(st.nu, st.freq))

    # ------------------------------------------------------------------ caching=True: HBM cache in front of the store
    def _sync_cache_hparams(self) -> None:
        c = self._cache
        c.learning_rate, c.eps, c.beta1, c.beta2, c.weight_decay = self.learning_rate, self.eps, self.beta1, self.beta2, self.weight_decay
        c._step, c._custom_score = self._step, self._custom_score

    def _forward_cached(self, indices: torch.Tensor, offsets: torch.Tensor, train: bool):
        """CACHING_PS forward (batched_dynamicemb_tables.py:694-706; the cache walk of batched_dynamicemb_function.py:298-556 with a
        `Storage` behind it): find in the HBM cache; misses are fetched from the store (`find`, padded value rows) or initialised,
        and inserted into the cache, whose evictions are written back (`insert`); a key the cache refuses (its bucket is full of
        rows this batch uses) is trained in a spill buffer and written back after the backward.  Every unique key ends up with ONE
        row address on the GPU, so gather and backward are the launches of the HBM-only module."""
        from .scored_hashtable import ScoreArg

        c = self._cache
        self._sync_cache_hparams()
        n, num_bags = indices.numel(), offsets.numel() - 1
        B = num_bags // self.feature_num
        T, dev = self.num_tables, self.device_
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        eb = torch.empty((), dtype=self.embedding_dtype).element_size()
        st = _StepCtx()
        st.offsets, st.num_keys, st.batch_size, st.num_bags = offsets, n, B, num_bags
        st.tids = st.slots = None
        st.pinned, st.event, st.scratch = False, None, None
        if pooled:
            out = torch.empty(B, self.total_D, dtype=self.output_dtype, device=dev)
            combiner = 0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=dev)
            combiner = -1
        rng = ext.get_table_range(offsets, self.feature_offsets)
        ukeys, st.rev, st.uoff, st.csr_cnt, st.csr_rank = ext.segmented_unique_csr(indices, rng, T)
        nu = int(st.uoff[-1].item())
        st.row_addr = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
        keep = []          # buffers the row addresses point into (alive until the gather has been queued / the backward has run)
        pins = []          # (slots, table ids) of the cache rows this training step holds: pinned until its backward
        self._drain_cache_orphans()
        if nu > 0:
            uk = ukeys[:nu].contiguous()
            tids = ext.expand_table_ids_cuda(st.uoff, nu)
            fp, fs, ip, isc, need_freq = c._scores(nu)
            if need_freq:
                fs = isc = st.csr_cnt[:nu].to(torch.int64)
            find = ScoreArg("score", None if fs is None else fs[:nu], fp)
            addr = st.row_addr[:nu]
            _, f0, s0 = c.table.lookup(uk, tids, find)
            hit0 = f0.nonzero().squeeze(1)
            if hit0.numel():
                addr[hit0] = ext.row_addresses(s0[hit0], tids[hit0], c.table_ptrs, c.table_value_dims, eb)
            miss = (~f0).nonzero().squeeze(1)
            if train and not miss.numel() and hit0.numel():     # (every key cached: the pins of the step are its hits)
                hs, ht = s0[hit0].contiguous(), tids[hit0].contiguous()
                c.table.increment_counter(hs, ht)
                pins.append((hs, ht))
            if miss.numel():
                k1, t1 = uk[miss].contiguous(), tids[miss].contiguous()
                fr1 = st.csr_cnt[:nu].to(torch.int64)[miss].contiguous() if need_freq else None
                if train:
                    nm, mkeys, midx, mtids, _, _, _, vals = _parse_find(self._storage.find(k1, t1, CopyMode.VALUE, fr1))
                    if vals.size(1) < self.max_V:
                        raise RuntimeError(f"Storage.find returned rows of {vals.size(1)} values, {self.max_V} expected")
                    if nm > 0:
                        midx = midx.to(torch.int64).contiguous()
                        self._init_padded_rows(vals, midx, mkeys, mtids if mtids is not None else t1[midx])
                    insn = ScoreArg("score", None if isc is None else isc[:nu][miss].contiguous(), ip)
                    # rows the batch found in the cache must not be evicted by the batch's own inserts -- nor, while this step
                    # waits for its backward, by a LATER forward's (a shared module called twice, gradient accumulation: the later
                    # step would evict a row whose address this step still holds, write the stale row back, and this step's
                    # backward would then update a slot that belongs to another key -- round-4 advisor finding).  The pins taken
                    # here are kept until _backward_cached (st.cache_pins); the inserted rows join them below.
                    if hit0.numel():
                        hs, ht = s0[hit0].contiguous(), tids[hit0].contiguous()
                        c.table.increment_counter(hs, ht)
                        pins.append((hs, ht))
                    idxn, h, ek, ei, es, et = c.table.insert_and_evict(k1, t1, insn)
                    if h:
                        ev = (ei >= 0).nonzero().squeeze(1)     # real evictions (negative entries mark refused inputs)
                        if ev.numel():   # the evicted rows go back to the store before anything overwrites them
                            e_k, e_s, e_sc, e_t = ek[ev].contiguous(), ei[ev].contiguous(), es[ev].contiguous(), et[ev].contiguous()
                            buf_ev = torch.zeros(ev.numel(), vals.size(1), dtype=self.embedding_dtype, device=dev)
                            ext.load_from_flat_table_value(c.table_ptrs, e_s, e_t, buf_ev, c.table_value_dims, c.table_emb_dims,
                                                           self.max_D, True)
                            self._storage.insert(e_k, e_t, buf_ev, e_sc.to(torch.int64))
                    ok = (idxn >= 0).nonzero().squeeze(1)
                    if ok.numel():
                        os_, ot_ = idxn[ok].contiguous(), t1[ok].contiguous()
                        c.table.increment_counter(os_, ot_)
                        pins.append((os_, ot_))
                        ext.store_to_flat_table_value(c.table_ptrs, idxn[ok].contiguous(), t1[ok].contiguous(), vals[ok].contiguous(),
                                                      c.table_value_dims, c.table_emb_dims, self.max_D, True)
                        addr[miss[ok]] = ext.row_addresses(idxn[ok].contiguous(), t1[ok].contiguous(), c.table_ptrs,
                                                           c.table_value_dims, eb)
                    bad = (idxn < 0).nonzero().squeeze(1)
                    if bad.numel():    # refused by the cache: flat rows in a spill buffer for this step, back to the store after it
                        # the spill buffer is a flat table of its own: the rows of table t, value_dims[t] wide, one after the other
                        # behind those of the tables before it (unique keys -- hence `bad` -- are table-major)
                        nb = bad.numel()
                        tb_ = t1[bad].contiguous()
                        cnt = torch.bincount(tb_, minlength=T)
                        first = torch.cumsum(cnt, 0) - cnt                                  # first spilled key of every table
                        rows = torch.arange(nb, dtype=torch.int64, device=dev) - first[tb_]  # row inside the table's region
                        elems = cnt * c.table_value_dims
                        spill = torch.zeros(nb * max(self.value_dims), dtype=self.embedding_dtype, device=dev)
                        sp_ptrs = (spill.data_ptr() + (torch.cumsum(elems, 0) - elems) * eb).contiguous()
                        ext.store_to_flat_table_value(sp_ptrs, rows, tb_, vals[bad].contiguous(), c.table_value_dims, c.table_emb_dims,
                                                      self.max_D, True)
                        addr[miss[bad]] = sp_ptrs[tb_] + rows * (c.table_value_dims[tb_] * eb)
                        sc_all = self._insert_scores(nu, fs[:nu] if need_freq else None)
                        st.scratch = (spill, sp_ptrs, rows, k1[bad].contiguous(), tb_, vals.size(1),
                                      None if sc_all is None else sc_all[miss][bad].contiguous())
                else:
                    nm, _, midx, _, _, _, _, vals = _parse_find(self._storage.find(k1, t1, CopyMode.EMBEDDING, None))
                    if nm > 0:
                        init_dense_rows(vals, midx, self._dynamicemb_options[0].eval_initializer_args)
                    addr[miss] = vals.data_ptr() + torch.arange(miss.numel(), dtype=torch.int64, device=dev) * (vals.stride(0) * eb)
                    keep.append(vals)
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims) and self.max_V % 4 == 0
        if pooled:
            check(lib().mi355_gather_pooled(None, 0, ptr(st.row_addr), dt(self.embedding_dtype), ptr(st.rev), n, ptr(offsets),
                                            num_bags, B, combiner, self.max_D, ptr(self.D_offsets_t), self.total_D, ptr(out),
                                            dt(out), int(al), stream()), "gather_pooled")
        elif n:
            check(lib().mi355_gather_rows(None, 0, ptr(st.row_addr), dt(self.embedding_dtype), ptr(st.rev), n, None, self.max_D,
                                          ptr(out), out.stride(0), dt(out), int(al), stream()), "gather_rows")
        del keep
        if not train:
            return out, None
        self._step += 1
        if pins:
            # a step that dies without its backward hands its pins to the module (released at the next forward, on a stream of
            # ours -- never from the garbage collector)
            import weakref

            cell = [pins]
            st.pin_cell = cell
            weakref.finalize(st, ExternalStorageTables._orphan_cache_step, weakref.ref(self), cell)
        return out, st

    @staticmethod
    def _orphan_cache_step(module_ref, cell) -> None:
        m = module_ref()
        if m is not None and cell[0] is not None:
            m._cache_orphans.append(cell[0])
            cell[0] = None

    def _drain_cache_orphans(self) -> None:
        orphans = self.__dict__.setdefault("_cache_orphans", [])
        while orphans:
            for slots, tids in orphans.pop():
                self._cache.table.decrement_counter(slots, tids)

    def _backward_cached(self, st, grads: torch.Tensor):
        if st is None:
            return
        c = self._cache
        self._sync_cache_hparams()
        c._backward_impl_inner(st, grads)
        cell = getattr(st, "pin_cell", None)
        if cell is not None and cell[0] is not None:       # the step's rows may be evicted again
            for slots, tids in cell[0]:
                c.table.decrement_counter(slots, tids)
            cell[0] = None
        self._iter_num = c._iter_num
        if st.scratch is not None:     # the rows the cache refused: trained in place in the spill buffer, now the store's again
            spill, sp_ptrs, rows, keys, tids, width, sc = st.scratch
            st.scratch = None
            buf = torch.zeros(keys.numel(), width, dtype=self.embedding_dtype, device=self.device_)
            ext.load_from_flat_table_value(sp_ptrs, rows, tids, buf, c.table_value_dims, c.table_emb_dims, self.max_D, True)
            self._storage.insert(keys, tids, buf, sc)
            del spill

    def flush(self) -> None:
        """write every row of the HBM cache back to the store (the rows stay cached)"""
        if self._cache is None:
            return
        c = self._cache
        for t in range(self.num_tables):
            d, s_ = self.dims[t], self.value_dims[t] - self.dims[t]
            for keys, rows, scores in c._export_table(t, 1 << 16):
                buf = torch.zeros(keys.numel(), self.max_V, dtype=self.embedding_dtype, device=self.device_)
                rows = rows.to(self.device_)
                buf[:, :d] = rows[:, :d]
                if s_:
                    buf[:, self.max_D:self.max_D + s_] = rows[:, d:d + s_]
                self._storage.insert(keys.to(self.device_), torch.full_like(keys, t, device=self.device_), buf,
                                     scores.to(self.device_).to(torch.int64))

    # ------------------------------------------------------------------ the rest of the surface
    def prefetch(self, *args, **kwargs):
        raise NotImplementedError("prefetch with an external storage")

    def reset_prefetch(self) -> None:
        return

    def train(self, mode: bool = True):
        return nn.Module.train(self, mode)

    def size(self, table_id: Optional[int] = None):
        """keys the store holds (with caching=True: after a flush(), which this call does not imply).  The `Storage` interface
        (types.py, as the reference's) counts over all tables: a per-table count is not something it offers"""
        if table_id is not None:
            raise NotImplementedError("size(table_id) with an external storage: Storage.size() counts all tables")
        return self._storage.size()

    @property
    def storage(self):
        return self._storage

    def _paths(self, save_dir: str, name: str, pg):
        rank, world = self._rank_world(pg)
        base = lambda item: os.path.join(save_dir, f"{name}_emb_{item}.rank_{rank}.world_size_{world}")  # noqa: E731
        return os.path.join(save_dir, f"{name}_opt_args.json"), base("keys"), base("values"), base("scores"), base("opt_values")

    def dump(self, save_dir: str, optim: bool = False, counter: bool = False, table_names=None, pg=None) -> None:
        self.flush()
        os.makedirs(save_dir, exist_ok=True)
        names = set(table_names if table_names is not None else self._table_names)
        for t, name in enumerate(self._table_names):
            if name not in names:
                continue
            meta, fk, fv, fs, fo = self._paths(save_dir, name, pg)
            with open(meta, "w") as f:
                json.dump(self._opt_args(), f)
            self._storage.dump(t, meta, fk, fv, fs, fo if optim else None, timestamp=ext.device_timestamp())

    def load(self, save_dir: str, optim: bool = False, counter: bool = False, table_names=None, pg=None) -> None:
        names = set(table_names if table_names is not None else self._table_names)
        for t, name in enumerate(self._table_names):
            if name not in names:
                continue
            meta, fk, fv, fs, fo = self._paths(save_dir, name, pg)
            self._storage.load(t, meta, fk, fv, fs, fo if optim else None, include_optim=optim, timestamp=ext.device_timestamp())
        if self._cache is not None:    # what the cache holds is stale now: the store is the truth again
            self._cache.table.reset()

    def export_keys_values(self, table_name: str, device: torch.device, batch_size: int = 65536):
        self.flush()
        return self._storage.export_keys_values(device, batch_size, self._table_names.index(table_name))
