"""`BatchedDynamicEmbeddingTablesV2` for MI355X: same constructor keywords, `forward(indices, offsets)`
contract and train/eval semantics as the reference module
(corelib/dynamicemb/dynamicemb/batched_dynamicemb_tables.py:452-1482, autograd function
batched_dynamicemb_function.py:1042-1300), on top of the sync-free gfx950 pipelines
(mi355_demb_forward / mi355_demb_backward).

What is implemented: HBM-only storage (the reference's `DynamicEmbStorage` in HBM_DIRECT mode),
pooling SUM / MEAN / NONE, mixed per-table dims for pooled mode, SGD / Adam / AdaGrad /
row-wise AdaGrad fused in the backward, score strategies TIMESTAMP / STEP / CUSTOMIZED / LFU,
train == eval for known keys, zeros for unknown keys in eval, first-touch insert + initialise in train.
Out of scope this round: cache / hybrid / host tiers, table growth (rehash), admission,
NO_EVICTION, dump / load.
"""
from __future__ import annotations

from itertools import accumulate
from typing import List, Optional

import torch
from torch import nn

import dynamicemb_extensions as ext
import mi355_native as N
from mi355_native import c_f, c_p, c_u64, check, dt, lib, ptr, stream

from .dynamicemb_config import (DynamicEmbCheckMode, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType,
                                get_optimizer_state_dim)
from .scored_hashtable import LinearBucketTable, ScoreSpec

_INIT_MODE = {DynamicEmbInitializerMode.UNIFORM: 0, DynamicEmbInitializerMode.NORMAL: 1,
              DynamicEmbInitializerMode.TRUNCATED_NORMAL: 2, DynamicEmbInitializerMode.CONSTANT: 3,
              DynamicEmbInitializerMode.DEBUG: 4}
_OPT_KIND = {"SGD": 1, "EXACT_SGD": 1, "ADAM": 2, "EXACT_ADAGRAD": 3, "EXACT_ROWWISE_ADAGRAD": 4}


class _StepCtx:
    """What one forward leaves for its backward (the reference's PrefetchState + autograd ctx)."""

    __slots__ = ("rev", "uoff", "tids", "slots", "row_addr", "offsets", "num_keys", "batch_size", "num_bags", "csr_cnt",
                 "csr_rank")


class _LookupFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, indices, offsets, dummy):
        out, step = module._forward_impl(indices, offsets, train=True)
        ctx.module = module
        ctx.step = step
        return out

    @staticmethod
    def backward(ctx, grads):
        ctx.module._backward_impl(ctx.step, grads)
        return None, None, None, None


class BatchedDynamicEmbeddingTablesV2(nn.Module):
    def __init__(self, table_options: List[DynamicEmbTableOptions], table_names: Optional[List[str]] = None,
                 feature_table_map: Optional[List[int]] = None, use_index_dedup: bool = False,
                 prefetch_pipeline: bool = False, pooling_mode: DynamicEmbPoolingMode = DynamicEmbPoolingMode.SUM,
                 output_dtype: torch.dtype = torch.float32, device: torch.device = None, enforce_hbm: bool = False,
                 bounds_check_mode=None, optimizer: EmbOptimType = EmbOptimType.SGD, stochastic_rounding: bool = True,
                 gradient_clipping: bool = False, max_gradient: float = 1.0, max_norm: float = 0.0,
                 learning_rate: float = 0.01, eps: float = 1.0e-8, initial_accumulator_value: float = 0.0,
                 momentum: float = 0.9, weight_decay: float = 0.0, weight_decay_mode=None, eta: float = 0.001,
                 beta1: float = 0.9, beta2: float = 0.999, counter_based_regularization=None,
                 cowclip_regularization=None, *args, **kwargs) -> None:
        super().__init__()
        assert len(table_options) >= 1
        opt0 = table_options[0]
        for o in table_options:
            assert opt0 == o, "All tables must match in grouped keys."
            if o.caching or o.external_storage is not None or o.admit_strategy is not None:
                raise NotImplementedError("cache / external storage / admission are 'next' rows (DESIGN.md)")
        self._dynamicemb_options = table_options
        self._table_names = table_names or [f"t{i}" for i in range(len(table_options))]
        self.pooling_mode = pooling_mode
        self.output_dtype = output_dtype
        self.use_index_dedup = use_index_dedup
        self.index_type = opt0.index_type or torch.int64
        self.embedding_dtype = opt0.embedding_dtype or torch.float32
        self.device_ = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dims: List[int] = [o.dim for o in table_options]
        if pooling_mode == DynamicEmbPoolingMode.NONE:
            assert all(d == self.dims[0] for d in self.dims), "Sequence mode requires uniform embedding dim"
        T_ = len(table_options)
        self.feature_table_map = feature_table_map if feature_table_map is not None else list(range(T_))
        assert all(any(t == m for m in self.feature_table_map) for t in range(T_)), "Each table must have at least one feature!"
        feature_dims = [self.dims[t] for t in self.feature_table_map]
        D_offsets = [0] + list(accumulate(feature_dims))
        self.total_D = D_offsets[-1]
        self.max_D = max(self.dims)
        self.mixed_D = self.max_D > min(self.dims)
        self.register_buffer("D_offsets_t", torch.tensor(D_offsets, device=self.device_, dtype=torch.int32)
                             if self.mixed_D else None)
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

        # optimizer (optimizer.py): row = [emb | state]
        if optimizer.name not in _OPT_KIND:
            raise ValueError(f"Not supported optimizer type: {optimizer}")
        self._opt_kind = _OPT_KIND[optimizer.name]
        self.optimizer_type = optimizer
        self.learning_rate, self.eps, self.beta1, self.beta2 = learning_rate, eps, beta1, beta2
        self.weight_decay = weight_decay
        self.initial_accumulator_value = initial_accumulator_value
        self._iter_num = 0
        state_dims = [get_optimizer_state_dim(optimizer, d, self.embedding_dtype) for d in self.dims]
        self.value_dims = [d + s for d, s in zip(self.dims, state_dims)]

        # score policy (batched_dynamicemb_tables.py `_create_score`, key_value_table.py:807-925)
        strat = opt0.score_strategy
        self._score_strategy = strat
        self._step = 0
        self._custom_score = 0
        if strat == DynamicEmbScoreStrategy.TIMESTAMP:
            pol = ext.ScorePolicy.GLOBAL_TIMER
        elif strat in (DynamicEmbScoreStrategy.STEP, DynamicEmbScoreStrategy.CUSTOMIZED):
            pol = ext.ScorePolicy.ASSIGN
        elif strat == DynamicEmbScoreStrategy.LFU:
            pol = ext.ScorePolicy.ACCUMULATE
        elif isinstance(strat, tuple):
            pol = ext.ScorePolicy.LRU_LFU
        else:
            raise NotImplementedError(f"score strategy {strat} is not supported yet (DESIGN.md)")
        self._policy = pol

        caps = [o.max_capacity for o in table_options]
        self.table = LinearBucketTable(caps, [ScoreSpec("score", pol)], key_type=torch.int64,
                                       bucket_capacity=opt0.bucket_capacity, device=self.device_)
        # flat value tables [capacity_t, value_dim_t] resident in HBM
        self.values = [torch.zeros(c, v, dtype=self.embedding_dtype, device=self.device_)
                       for c, v in zip(self.table.per_table_capacity_, self.value_dims)]
        self.table_ptrs = torch.tensor([v.data_ptr() for v in self.values], dtype=torch.int64, device=self.device_)
        self.table_value_dims = torch.tensor(self.value_dims, dtype=torch.int64, device=self.device_)
        self.table_emb_dims = torch.tensor(self.dims, dtype=torch.int64, device=self.device_)
        self.initializer_args = opt0.initializer_args
        self._seed = 1234
        self._score_buf = None
        self._empty_tensor = nn.Parameter(torch.empty(10, requires_grad=True, device=self.device_,
                                                      dtype=self.embedding_dtype))
        self._pin = False  # ref-counter pinning is only needed when prefetch runs ahead of backward

    # ---------------------------------------------------------------------------------- helpers
    def set_score(self, score: int) -> None:
        self._custom_score = int(score)

    def set_learning_rate(self, lr: float) -> None:
        self.learning_rate = lr

    @property
    def optimizer_step(self) -> int:
        return self._iter_num

    def _init_params(self):
        a = self.initializer_args
        m = _INIT_MODE[a.mode]
        if a.mode == DynamicEmbInitializerMode.UNIFORM:
            lo = a.lower if a.lower is not None else 0.0
            hi = a.upper if a.upper is not None else 1.0
            return m, (lo, hi, 0.0, 0.0)
        if a.mode == DynamicEmbInitializerMode.NORMAL:
            return m, (a.mean, a.std_dev, 0.0, 0.0)
        if a.mode == DynamicEmbInitializerMode.TRUNCATED_NORMAL:
            return m, (a.mean, a.std_dev, a.lower if a.lower is not None else -2.0, a.upper if a.upper is not None else 2.0)
        if a.mode == DynamicEmbInitializerMode.CONSTANT:
            return m, (a.value, 0.0, 0.0, 0.0)
        return m, (0.0, 0.0, 0.0, 0.0)

    def _scores(self, n: int):
        """(find_policy, find_scores, insert_policy, insert_scores, freq?) for this step."""
        P = ext.ScorePolicy
        s = self._score_strategy
        if s == DynamicEmbScoreStrategy.TIMESTAMP:
            return P.GLOBAL_TIMER, None, P.GLOBAL_TIMER, None, False
        if s in (DynamicEmbScoreStrategy.STEP, DynamicEmbScoreStrategy.CUSTOMIZED):
            val = self._step if s == DynamicEmbScoreStrategy.STEP else self._custom_score
            if self._score_buf is None or self._score_buf.numel() < n:
                self._score_buf = torch.empty(max(n, 1), dtype=torch.int64, device=self.device_)
            self._score_buf.fill_(val)
            return P.ASSIGN, self._score_buf, P.ASSIGN, self._score_buf, False
        if s == DynamicEmbScoreStrategy.LFU:
            return P.ACCUMULATE, None, P.ASSIGN, None, True
        return P.LRU_LFU, None, P.LRU_LFU, None, True

    # ---------------------------------------------------------------------------------- forward
    def _forward_impl(self, indices: torch.Tensor, offsets: torch.Tensor, train: bool):
        indices = indices.contiguous()
        offsets = offsets.to(torch.int64).contiguous()
        if indices.dtype != torch.int64:
            indices = indices.to(torch.int64)
        n = indices.numel()
        num_bags = offsets.numel() - 1
        B = num_bags // self.feature_num
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        dev = self.device_
        T = self.num_tables
        st = _StepCtx()
        st.rev = torch.empty(n, dtype=torch.int64, device=dev)
        st.uoff = torch.empty(T + 1, dtype=torch.int64, device=dev)
        st.tids = torch.empty(n, dtype=torch.int64, device=dev)
        st.slots = torch.empty(n, dtype=torch.int64, device=dev)
        st.row_addr = torch.empty(n, dtype=torch.int64, device=dev)
        # CSR ingredients for the backward (occurrences per unique row, rank of every key inside its row's list)
        st.csr_cnt = torch.empty(n, dtype=torch.int32, device=dev) if train else None
        st.csr_rank = torch.empty(n, dtype=torch.int32, device=dev) if train else None
        st.offsets, st.num_keys, st.batch_size, st.num_bags = offsets, n, B, num_bags
        if pooled:
            out = torch.empty(B, self.total_D, dtype=self.output_dtype, device=dev)
            combiner = 0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=dev)
            combiner = -1
        fp, fs, ip, isc, need_freq = self._scores(n)
        freq = torch.empty(n, dtype=torch.int64, device=dev) if need_freq else None
        mode, p = self._init_params()
        state_init = self.initial_accumulator_value
        ws = torch.empty(lib().mi355_demb_forward_workspace_bytes(n, T), dtype=torch.uint8, device=dev)
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims)
        tb = self.table
        check(lib().mi355_demb_forward(
            ptr(tb.table_storage_), ptr(tb.table_bucket_offsets_), tb.bucket_capacity_, tb.num_scores_,
            ptr(tb.bucket_sizes), ptr(tb._ref_counter), tb._ref_counter.numel(),
            ptr(self.table_ptrs), ptr(self.table_value_dims), ptr(self.table_emb_dims), dt(self.embedding_dtype),
            self.max_D, max(self.value_dims),
            ptr(indices), n, ptr(offsets), num_bags, B, ptr(self.feature_offsets), T,
            int(train), int(fp), ptr(fs), int(ip), ptr(isc), c_u64(ext.TIMER_OVERRIDE), int(self._pin and train),
            mode, c_f(p[0]), c_f(p[1]), c_f(p[2]), c_f(p[3]), c_u64(self._seed), c_f(state_init),
            combiner, ptr(self.D_offsets_t), self.total_D, ptr(out), dt(out), int(al),
            ptr(st.rev), ptr(st.uoff), ptr(st.tids), ptr(st.slots), ptr(st.row_addr), ptr(freq),
            ptr(st.csr_cnt), ptr(st.csr_rank), ptr(ws), ws.numel(), stream()), "demb_forward")
        if train:
            self._step += 1
            if self._dynamicemb_options[0].safe_check_mode != DynamicEmbCheckMode.IGNORE:
                self._safe_check(st)
        return out, st

    def _safe_check(self, st):
        nu = int(st.uoff[-1].item())
        failed = int((st.slots[:nu] < 0).sum().item())
        if failed:
            msg = f"DynamicEmb: {failed} of {nu} unique keys could not be inserted (bucket full of pinned/locked slots)"
            if self._dynamicemb_options[0].safe_check_mode == DynamicEmbCheckMode.ERROR:
                raise RuntimeError(msg)
            import warnings

            warnings.warn(msg)

    def _backward_impl(self, st, grads: torch.Tensor):
        grads = grads.contiguous()
        self._iter_num += 1
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        combiner = -1 if not pooled else (0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1)
        dim = self.max_D
        ws = torch.empty(lib().mi355_demb_backward_workspace_bytes(st.num_keys, dim), dtype=torch.uint8, device=grads.device)
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims) and grads.stride(0) % 4 == 0
        tb = self.table
        check(lib().mi355_demb_backward(
            ptr(st.rev), st.num_keys, ptr(st.uoff), self.num_tables, ptr(st.offsets), st.num_bags, st.batch_size,
            ptr(grads), grads.stride(0), dt(grads), ptr(self.D_offsets_t), dim, combiner, ptr(st.row_addr),
            dt(self.embedding_dtype), self._opt_kind, c_f(self.learning_rate), c_f(self.beta1), c_f(self.beta2),
            c_f(self.eps), c_f(self.weight_decay), self._iter_num, -1, 1, int(al),
            ptr(tb._ref_counter), tb._ref_counter.numel(), ptr(st.slots), ptr(st.tids), ptr(tb.table_bucket_offsets_),
            tb.bucket_capacity_, int(self._pin), ptr(st.csr_cnt), ptr(st.csr_rank), ptr(ws), ws.numel(), stream()),
            "demb_backward")

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, per_sample_weights=None,
                feature_requires_grad=None, batch_size_per_feature_per_rank=None, total_unique_indices=None):
        if per_sample_weights is not None:
            raise NotImplementedError("per_sample_weights is not supported (nor by the reference's kernels)")
        if self.training and torch.is_grad_enabled():
            return _LookupFunction.apply(self, indices, offsets, self._empty_tensor)
        out, _ = self._forward_impl(indices, offsets, train=False)
        return out

    # ---------------------------------------------------------------------------------- inspection
    def size(self, table_id: Optional[int] = None):
        return self.table.size(table_id)

    def lookup_rows(self, keys: torch.Tensor, table_id: int = 0):
        """(found, rows [n, value_dim]) of `keys` -- test / debugging helper (CONST lookup)."""
        from .scored_hashtable import ScoreArg

        tids = torch.full_like(keys, table_id)
        _, found, idx = self.table.lookup(keys, tids, ScoreArg("score", None, ext.ScorePolicy.CONST))
        rows = self.values[table_id][idx.clamp(min=0)]
        rows[~found] = 0
        return found, rows
