"""`BatchedDynamicEmbeddingTablesV2` for MI355X: same constructor keywords, `forward(indices, offsets)`
contract and train/eval semantics as the reference module
(corelib/dynamicemb/dynamicemb/batched_dynamicemb_tables.py:452-1482, autograd function
batched_dynamicemb_function.py:1042-1300), on top of the sync-free gfx950 pipelines
(mi355_demb_forward / mi355_demb_backward).

What is implemented: HBM / host / hybrid / promoting-cache storage tiers, pooling SUM / MEAN / NONE, mixed per-table
dims for pooled mode, SGD / Adam / AdaGrad / row-wise AdaGrad fused in the backward, score strategies TIMESTAMP /
STEP / CUSTOMIZED / LFU, train == eval for known keys, zeros for unknown keys in eval, first-touch insert + initialise
in train, prefetch(), dump / load / export, frequency admission (`admit_strategy` + `admission_counter`).
Table growth by rehash (init_capacity -> max_capacity, VMM value buffers) for HBM storage.  Not built: NO_EVICTION,
external storage.
"""
from __future__ import annotations

import ctypes
import os
from itertools import accumulate
from typing import List, Optional

import torch
from torch import nn

import dynamicemb_extensions as ext
import mi355_native as N
from mi355_native import c_f, c_p, c_u64, check, current_torch_stream, dt, lib, ptr, stream
from mi355_native import _DT as _DT_CODE, _cur_device, _raw_stream

from .dynamicemb_config import (DynamicEmbCheckMode, DynamicEmbInitializerMode, DynamicEmbPoolingMode,
                                DynamicEmbScoreStrategy, DynamicEmbTableOptions, EmbOptimType,
                                get_optimizer_state_dim)
from .scored_hashtable import LinearBucketTable, ScoreSpec

_INIT_MODE = {DynamicEmbInitializerMode.UNIFORM: 0, DynamicEmbInitializerMode.NORMAL: 1,
              DynamicEmbInitializerMode.TRUNCATED_NORMAL: 2, DynamicEmbInitializerMode.CONSTANT: 3,
              DynamicEmbInitializerMode.DEBUG: 4}
_OPT_KIND = {"SGD": 1, "EXACT_SGD": 1, "ADAM": 2, "EXACT_ADAGRAD": 3, "EXACT_ROWWISE_ADAGRAD": 4}


def init_dense_rows(buffer: torch.Tensor, indices: torch.Tensor, args) -> None:
    """rows `indices` of the dense `buffer` <- initializer `args` (the reference's BaseDynamicEmbInitializer call on a
    dense buffer, initializer.py); counter-based generators keyed by the row index"""
    m = _INIT_MODE[args.mode]
    if indices.numel() == 0:
        return
    if args.mode == DynamicEmbInitializerMode.UNIFORM:
        p = (args.lower if args.lower is not None else 0.0, args.upper if args.upper is not None else 1.0, 0.0, 0.0)
    elif args.mode == DynamicEmbInitializerMode.NORMAL:
        p = (args.mean, args.std_dev, 0.0, 0.0)
    elif args.mode == DynamicEmbInitializerMode.TRUNCATED_NORMAL:
        p = (args.mean, args.std_dev, args.lower if args.lower is not None else -2.0, args.upper if args.upper is not None else 2.0)
    elif args.mode == DynamicEmbInitializerMode.CONSTANT:
        p = (args.value, 0.0, 0.0, 0.0)
    else:
        p = (0.0, 0.0, 0.0, 0.0)
    eb = buffer.element_size()
    addr = buffer.data_ptr() + indices.to(torch.int64) * (buffer.stride(0) * eb)
    ext.init_rows(m, p, 1234, 0.0, indices.to(torch.int64).contiguous(), addr.contiguous(), buffer.dtype, buffer.size(1),
                  buffer.size(1))


class _StepCtx:
    """What one forward leaves for its backward (the reference's PrefetchState + autograd ctx)."""

    __slots__ = ("rev", "uoff", "tids", "slots", "row_addr", "offsets", "num_keys", "batch_size", "num_bags", "csr_cnt",
                 "csr_rank", "pinned", "event", "indices", "bwd_ws", "ring", "token", "tier_pins", "fwd_addr", "scratch",
                 "pin_cell", "__weakref__")

    def release_ring(self):
        """hand the early-CSR ring slot back (after the backward, or when the step is dropped without one)"""
        r = getattr(self, "ring", None)
        if r is not None:
            r[0][r[1]] = False
            self.ring = None

    def __del__(self):
        self.release_ring()


def _al256(x: int) -> int:
    return (x + 255) // 256 * 256


class _FusedStep:
    """Step context of the fused forward (mi355_demb_forward_fused): every per-step array lives in ONE buffer (a slot of
    the module's ring, or a fresh allocation); the arrays the side stream produces (unique numbering, CSR) are joined
    before anybody looks at them."""

    _FIELDS = (("rev", 8, torch.int64), ("tids", 8, torch.int64), ("slots", 8, torch.int64), ("row_addr", 8, torch.int64),
               ("freq", 8, torch.int64), ("csr_cnt", 4, torch.int32), ("csr_rank", 4, torch.int32))

    def __init__(self, module, buf, n, T, fwd_ws_bytes, bwd_ws_bytes):
        self.module, self.buf, self.num_keys, self.T = module, buf, n, T
        off = 0
        self.off = {}
        for name, eb, _ in self._FIELDS:
            self.off[name] = off
            off += _al256(eb * max(n, 1))
        self.off["uoff"] = off
        off += _al256(8 * (T + 1))
        self.off["fwd_ws"] = off
        off += _al256(fwd_ws_bytes)
        self.off["bwd_ws"] = off
        off += _al256(bwd_ws_bytes)
        self.total = off
        self.fwd_ws_bytes, self.bwd_ws_bytes = fwd_ws_bytes, bwd_ws_bytes
        self.token = -1        # side-stream join point (-1: nothing to join)
        self.ring = None
        self.pinned = False
        self.event = None
        self.indices = None
        self.offsets = None
        self.prepared = 0
        self.lazy = False      # reverse indices / ranks not produced yet (mi355_demb_fused_materialize on first use)
        self.epoch = 0         # path (c): the step's overflow notice has not been read yet (settle())
        self.rerun_args = None

    def settle(self):
        """A path-(c) step gave every slot-range partition a fixed record list.  Before anybody uses its CSR or unique numbering
        the library is asked whether every list held its records (one read of pinned memory, written by the step's partition
        kernel); a flooded step is regrouped on the per-slot-counter path first.  No step ever skips its update (reference:
        batched_dynamicemb_function.py:1042-1191, the unique op serves any key stream)."""
        ep = self.epoch
        if ep:
            self.epoch = 0
            f = lib().mi355_demb_fused_step_flooded(ep, 20000)
            if f < 0:
                raise RuntimeError("fused forward: the step never reported its partition state (GPU stuck?)")
            if f == 3:
                raise RuntimeError("fused forward: a gather block gave up waiting for the partition block of its own launch "
                                   "(blocks were not dispatched in id order); the step's output lacks rows")
            if f:
                self.module._rerun_step(self, ep)

    def materialize(self):
        """per-occurrence outputs of a forward whose partition blocks wrote the CSR themselves (join_token <= -2)"""
        if self.epoch:
            self.settle()
        if self.lazy:
            self.lazy = False
            check(lib().mi355_demb_fused_materialize(self.p("fwd_ws"), self.fwd_ws_bytes, self.num_keys, self.T, self.p("row_addr"),
                                                     self.p("rev"), self.p("csr_rank"), stream()), "fused_materialize")

    @staticmethod
    def nbytes(n, T, fwd_ws_bytes, bwd_ws_bytes):
        return (sum(_al256(eb * max(n, 1)) for _, eb, _ in _FusedStep._FIELDS) + _al256(8 * (T + 1)) + _al256(fwd_ws_bytes)
                + _al256(bwd_ws_bytes))

    def p(self, name):
        return c_p(self.buf.data_ptr() + self.off[name])

    def join(self):
        """make the current stream wait for the side-stream half of this step's forward (and settle a path-(c) step)"""
        if self.epoch:
            self.settle()
        if self.token >= 0:
            check(lib().mi355_side_join(self.token, stream()), "side join")
            self.token = -1

    def _view(self, name):
        self.join()
        if name in ("rev", "csr_rank"):
            self.materialize()
        for nm, eb, dtp in self._FIELDS:
            if nm == name:
                o = self.off[nm]
                return self.buf[o:o + eb * self.num_keys].view(dtp)
        raise AttributeError(name)

    rev = property(lambda self: self._view("rev"))
    tids = property(lambda self: self._view("tids"))
    slots = property(lambda self: self._view("slots"))
    row_addr = property(lambda self: self._view("row_addr"))
    csr_cnt = property(lambda self: self._view("csr_cnt"))
    csr_rank = property(lambda self: self._view("csr_rank"))

    @property
    def uoff(self):
        self.join()
        o = self.off["uoff"]
        return self.buf[o:o + 8 * (self.T + 1)].view(torch.int64)

    def release_ring(self):
        r = self.ring
        if r is not None:
            r[0][r[1]] = False
            self.ring = None

    def __del__(self):
        self.release_ring()


class _PlanStep(_FusedStep):
    """Step context of the pre-bound training step (mi355_demb_plan_forward): the same single buffer as _FusedStep, but nothing is
    computed on the hot path -- the array offsets come from the library's own layout function the first time somebody asks."""

    def __init__(self, module, buf, n, T, ring, offsets, B, num_bags):   # (no super().__init__: that is the eager layout)
        self.module, self.buf, self.num_keys, self.T = module, buf, n, T
        self.ring = ring
        self.offsets, self.batch_size, self.num_bags = offsets, B, num_bags
        self.token = -1
        self.pinned = False
        self.event = None
        self.indices = None
        self.prepared = 0
        self.lazy = False
        self.epoch = 0
        self.rerun_args = None
        self.plan_step = True

    def _layout(self):
        lay = (ctypes.c_int64 * 13)()
        lib().mi355_demb_step_layout(self.num_keys, self.T, self.module.max_D, 1, lay)
        names = [f[0] for f in self._FIELDS] + ["uoff", "fwd_ws", "bwd_ws"]
        self.__dict__["off"] = {nm: int(lay[i]) for i, nm in enumerate(names)}
        self.__dict__["total"], self.__dict__["fwd_ws_bytes"], self.__dict__["bwd_ws_bytes"] = int(lay[10]), int(lay[11]), int(lay[12])

    def __getattr__(self, name):          # only reached for attributes that are not set yet
        if name in ("off", "total", "fwd_ws_bytes", "bwd_ws_bytes"):
            self._layout()
            return self.__dict__[name]
        raise AttributeError(name)


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
    def __new__(cls, table_options=None, *args, **kwargs):
        # tables backed by a user-supplied store (options.external_storage) are the subclass of external_storage.py
        if cls is BatchedDynamicEmbeddingTablesV2 and table_options and any(o.external_storage is not None for o in table_options):
            from .external_storage import ExternalStorageTables

            return super().__new__(ExternalStorageTables)
        return super().__new__(cls)

    def __init__(self, table_options: List[DynamicEmbTableOptions], table_names: Optional[List[str]] = None,
                 feature_table_map: Optional[List[int]] = None, use_index_dedup: bool = False,
                 prefetch_pipeline: bool = False, pooling_mode: DynamicEmbPoolingMode = DynamicEmbPoolingMode.SUM,
                 output_dtype: torch.dtype = torch.float32, device: torch.device = None, enforce_hbm: bool = False,
                 bounds_check_mode=None, optimizer: EmbOptimType = EmbOptimType.SGD, stochastic_rounding: bool = True,
                 gradient_clipping: bool = False, max_gradient: float = 1.0, max_norm: float = 0.0,
                 learning_rate: float = 0.01, eps: float = 1.0e-8, initial_accumulator_value: float = 0.0,
                 momentum: float = 0.9, weight_decay: float = 0.0, weight_decay_mode=None, eta: float = 0.001,
                 beta1: float = 0.9, beta2: float = 0.999, counter_based_regularization=None,
                 cowclip_regularization=None, storage_mode: Optional[str] = None, *args, **kwargs) -> None:
        super().__init__()
        assert len(table_options) >= 1
        opt0 = table_options[0]
        for o in table_options:
            assert opt0 == o, "All tables must match in grouped keys."
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
        # table growth (key_value_table.py:440-666): start at init_capacity, double by rehash while
        # (size + incoming) / capacity > max_load_factor, up to max_capacity.  HBM storage only; the value buffers are
        # VMM tensors reserved for max_capacity rows, so their base address survives every growth step.
        self._max_caps = list(caps)
        self._growth = False
        if (storage_mode in (None, "hbm") and all(o.init_capacity is not None and 0 < o.init_capacity < o.max_capacity
                                                  for o in table_options)
                and not (0 < opt0.local_hbm_for_values < sum(c * v * torch.empty((), dtype=self.embedding_dtype).element_size()
                                                             for c, v in zip(caps, self.value_dims)))):
            self._growth = True
            caps = [o.init_capacity for o in table_options]
        # ---- storage tiers (batched_dynamicemb_tables.py:637-787, key_value_table.py:1522-2403) -------------------
        #  hbm    : hash table + rows in HBM (DynamicEmbStorage on device)
        #  host   : hash table + rows in pinned host memory, driven by the same kernels over the host link
        #  hybrid : HBM tier limited to `local_hbm_for_values` bytes + host tier (HybridStorage): a key lives in exactly
        #           one tier, new keys enter the HBM tier, what it evicts spills to the host tier.  The gather / backward
        #           kernels take row ADDRESSES, so rows of both tiers are mixed freely in one launch -- nothing is staged.
        # Deviation: `local_hbm_for_values == 0` means "no limit" here (288 GB of HBM per GPU); the reference reads 0
        # as host-only.  `storage_mode="host"` selects that explicitly.
        elem = torch.empty((), dtype=self.embedding_dtype).element_size()
        row_bytes = [v * elem for v in self.value_dims]
        total_bytes = sum(c * b for c, b in zip(caps, row_bytes))
        if storage_mode is None:
            storage_mode = "hybrid" if 0 < opt0.local_hbm_for_values < total_bytes else "hbm"
            if opt0.caching and storage_mode == "hybrid":
                storage_mode = "cache"
        assert storage_mode in ("hbm", "host", "hybrid", "cache")
        #  cache  : `caching=True` (DynamicEmbCache + backing storage, key_value_table.py:1522-1647): the hybrid layout
        #           plus PROMOTION -- a key found in the host tier moves into the HBM tier (whose evictions move down), so
        #           the HBM tier converges to the hot set.  A key still lives in exactly one tier, hence flush() is a no-op.
        self._promote = storage_mode == "cache"
        if storage_mode == "cache":
            storage_mode = "hybrid"
        self.storage_mode = storage_mode
        C = opt0.bucket_capacity

        self._vmm = None

        def _values(capacities, host):
            if self._growth and not host:
                self._vmm = [ext.VMMTensor(c * v, self.embedding_dtype, self.device_.index or 0, reserve_numel=m * v)
                             for c, v, m in zip(capacities, self.value_dims, self._max_caps)]
                return [b.data().view(c, v) for b, c, v in zip(self._vmm, capacities, self.value_dims)]
            if host:
                return [torch.zeros(c, v, dtype=self.embedding_dtype).pin_memory() for c, v in zip(capacities, self.value_dims)]
            return [torch.zeros(c, v, dtype=self.embedding_dtype, device=self.device_) for c, v in zip(capacities, self.value_dims)]

        if storage_mode == "hybrid":
            share = [opt0.local_hbm_for_values * (c * b) // max(total_bytes, 1) for c, b in zip(caps, row_bytes)]
            hbm_caps = [max(C, (sh // b) // C * C) for sh, b in zip(share, row_bytes)]
            self.table = LinearBucketTable(hbm_caps, [ScoreSpec("score", pol)], key_type=torch.int64, bucket_capacity=C,
                                           device=self.device_)
            self.table_host = LinearBucketTable(caps, [ScoreSpec("score", pol)], key_type=torch.int64, bucket_capacity=C,
                                                device=self.device_, host=True)
            self.values = _values(self.table.per_table_capacity_, False)
            self.values_host = _values(self.table_host.per_table_capacity_, True)
            self.table_ptrs_host = torch.tensor([v.data_ptr() for v in self.values_host], dtype=torch.int64, device=self.device_)
        else:
            host = storage_mode == "host"
            self.table = LinearBucketTable(caps, [ScoreSpec("score", pol)], key_type=torch.int64, bucket_capacity=C,
                                           device=self.device_, host=host)
            self.table_host = None
            # flat value tables [capacity_t, value_dim_t]
            self.values = _values(self.table.per_table_capacity_, host)
        # ---- admission (batched_dynamicemb_tables.py:526,624,798-812): one fused counter table for all logical tables
        self._admit_strategy = opt0.admit_strategy
        counters = [o.admission_counter for o in table_options]
        if all(c is None for c in counters):
            self._admission_counter = None
        else:
            assert all(c is not None for c in counters), "All tables must either have or not have an admission counter"
            from .embedding_admission import MultiTableKVCounter

            self._admission_counter = MultiTableKVCounter(counters, device=self.device_)
        if self._admit_strategy is not None:
            if self._admission_counter is None:
                raise ValueError("admit_strategy needs an admission_counter (KVCounter) per table")
        self.table_ptrs = torch.tensor([v.data_ptr() for v in self.values], dtype=torch.int64, device=self.device_)
        self.table_value_dims = torch.tensor(self.value_dims, dtype=torch.int64, device=self.device_)
        self.table_emb_dims = torch.tensor(self.dims, dtype=torch.int64, device=self.device_)
        self.initializer_args = opt0.initializer_args
        self._seed = 1234
        self._score_buf = None
        self._empty_tensor = nn.Parameter(torch.empty(10, requires_grad=True, device=self.device_,
                                                      dtype=self.embedding_dtype))
        self._pin = False  # ref-counter pinning is only needed when prefetch runs ahead of backward
        self._early_csr = True   # (per-op path) build the backward's CSR on the side stream under the forward
        # fused index stage (csrc/fused_fwd.hip): HBM storage, no admission, 32-bit slot ids, not the deterministic mode
        self._fused = (os.environ.get("MI355_FUSED", "1") != "0" and storage_mode == "hbm" and self._admit_strategy is None
                       and T_ <= 128 and self.table.capacity_ < (1 << 31) - 512
                       and os.environ.get("DEMB_DETERMINISM_MODE", "") in ("", "0"))
        self._fused_aux = None
        import weakref
        self._live_steps = weakref.WeakSet()   # forward contexts whose backward is still to come
        self._fused_side = False   # (the fused forward forks nothing: measured slower back to back)
        self._plan = None
        # the pre-bound step (_plan_forward): the steady-state configuration only -- everything else keeps the general path
        self._plan_ok = (self._fused and not self._growth and not self._fused_side and os.environ.get("MI355_PLAN", "1") != "0"
                         and opt0.safe_check_mode == DynamicEmbCheckMode.IGNORE and _raw_stream is not None and _cur_device is not None)
        self._step_ring = [None] * 4
        self._bwd_ring = [None] * 4
        self._bwd_busy = [False] * 4
        self._bwd_ring_next = 0
        from collections import deque

        self._prefetch_states = deque()
        # round 6: prefetch on the partitioned index path (mi355_demb_plan_stage).  Scores of the steps in flight (forward or
        # prefetch issued, backward not yet): an eviction spares every slot that scores at least the smallest of them
        self._inflight = weakref.WeakKeyDictionary()
        self._pf_c_used = False
        self._clock0 = None
        self._tier_prefetched = 0      # prefetched batches of the tier / admission paths whose backward has not run yet
        self._orphan_pins = []         # pins of steps that died without a backward (released at the next call)
        self._grow_deferred = False    # a growth that had to wait for live / prefetched steps (retried after a backward)

    # ---------------------------------------------------------------------------------- helpers
    def set_score(self, score: int) -> None:
        self._custom_score = int(score)

    def set_learning_rate(self, lr: float) -> None:
        self.learning_rate = lr

    @property
    def optimizer_step(self) -> int:
        return self._iter_num

    def _init_params(self, a=None):
        a = a if a is not None else self.initializer_args
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
    def _forward_impl(self, indices: torch.Tensor, offsets: torch.Tensor, train: bool, prefetch_only: bool = False):
        if (train and self._plan_ok and not prefetch_only and not self._pin and not self._prefetch_states and not self._orphan_pins
                and indices.dtype is torch.int64 and offsets.dtype is torch.int64 and indices.is_contiguous()
                and offsets.is_contiguous() and indices.is_cuda):
            return self._plan_forward(indices, offsets)
        if self._orphan_pins:
            self._drain_orphan_pins()
        out, st = self._forward_impl_inner(indices, offsets, train, prefetch_only)
        if train and st is not None:
            # steps whose backward has not been issued yet hold slots / row addresses of the CURRENT table: growth (a rehash
            # moves every row) waits for them (_maybe_grow); a step dropped without a backward leaves the set when it dies
            self._live_steps.add(st)
        return out, st

    def _forward_impl_inner(self, indices: torch.Tensor, offsets: torch.Tensor, train: bool, prefetch_only: bool = False):
        indices = indices.contiguous()
        offsets = offsets.to(torch.int64).contiguous()
        if indices.dtype != torch.int64:
            indices = indices.to(torch.int64)
        if not prefetch_only and train and self._prefetch_states:
            st = self._prefetch_states.popleft()
            if st.num_keys != indices.numel() or st.num_bags != offsets.numel() - 1:
                raise RuntimeError("forward() received a batch that was not the oldest prefetched one")
            if getattr(st, "staged", False):
                return self._plan_gather_staged(st), st
            return self._gather_prefetched(st), st
        if self.storage_mode == "hybrid":
            return self._forward_hybrid(indices, offsets, train, prefetch_only)
        if self._admit_strategy is not None and train:
            return self._forward_admission(indices, offsets, prefetch_only)
        if self._growth and train:
            self._maybe_grow(indices.numel())
        if self._fused and self.table.capacity_ < (1 << 31) - 512:   # (a table grown past 32-bit slot ids takes the per-op chain)
            return self._forward_fused(indices, offsets, train, prefetch_only)
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
        st.pinned = bool((self._pin or prefetch_only) and train)
        st.event = None
        if prefetch_only:
            out, combiner = None, -2
        elif pooled:
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
        # early CSR: the backward's key grouping runs on the library's side stream under this forward's lookup / gather
        # kernels; the buffer travels to the backward in the step context.  Not under stream capture (the side stream
        # joins in the backward, a forward captured alone would leave it dangling).
        st.bwd_ws = None
        if (train and n > 0 and self._early_csr and not prefetch_only and torch.is_grad_enabled()
                and not torch.cuda.is_current_stream_capturing()):
            # a small ring of module-owned buffers (never handed back to the allocator, so a step that never gets its
            # backward cannot have its buffer reused under the side-stream kernels; writers are ordered by the side stream)
            # A slot is busy from its forward until its backward (or until the step context is dropped); when more steps
            # are outstanding than the ring holds, the extra ones simply group in their backward as before.
            slot = self._bwd_ring_next % len(self._bwd_ring)
            if not self._bwd_busy[slot]:
                self._bwd_ring_next += 1
                need = lib().mi355_demb_backward_workspace_bytes(n, self.max_D)
                buf = self._bwd_ring[slot]
                if buf is None or buf.numel() < need:
                    buf = self._bwd_ring[slot] = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
                self._bwd_busy[slot] = True
                st.ring = (self._bwd_busy, slot)
                st.bwd_ws = buf
        tok = ctypes.c_int(-1)
        check(lib().mi355_demb_forward(
            ptr(tb.table_storage_), ptr(tb.table_bucket_offsets_), tb.bucket_capacity_, tb.num_scores_,
            ptr(tb.bucket_sizes), ptr(tb._ref_counter), tb._ref_counter.numel(),
            ptr(self.table_ptrs), ptr(self.table_value_dims), ptr(self.table_emb_dims), dt(self.embedding_dtype),
            self.max_D, max(self.value_dims),
            ptr(indices), n, ptr(offsets), num_bags, B, ptr(self.feature_offsets), T,
            int(train), int(fp), ptr(fs), int(ip), ptr(isc), c_u64(ext.TIMER_OVERRIDE), int(st.pinned),
            mode, c_f(p[0]), c_f(p[1]), c_f(p[2]), c_f(p[3]), c_u64(self._seed), c_f(state_init),
            combiner, ptr(self.D_offsets_t), self.total_D, ptr(out), dt(out) if out is not None else 0, int(al),
            ptr(st.rev), ptr(st.uoff), ptr(st.tids), ptr(st.slots), ptr(st.row_addr), ptr(freq),
            ptr(st.csr_cnt), ptr(st.csr_rank), ptr(st.bwd_ws), st.bwd_ws.numel() if st.bwd_ws is not None else 0,
            ctypes.byref(tok), ptr(ws), ws.numel(), stream()), "demb_forward")
        st.token = tok.value
        if st.token >= 0:
            # the side-stream kernels read these allocator-owned tensors: a step that is dropped without its backward must
            # not hand them back to the allocator while they are in use there
            side = self._side_stream_obj()
            for t_ in (st.rev, st.csr_cnt, st.csr_rank, st.uoff, offsets):
                if t_ is not None:
                    t_.record_stream(side)
        if train:
            self._step += 1
            if self._dynamicemb_options[0].safe_check_mode != DynamicEmbCheckMode.IGNORE:
                self._safe_check(st)
        return out, st

    def _side_stream_obj(self):
        """the library's side stream (mi355_early_csr_stream) as a torch stream, for record_stream"""
        so = getattr(self, "_side_stream_cache", None)
        if so is None:
            L = lib()
            L.mi355_early_csr_stream.restype = ctypes.c_void_p
            so = self._side_stream_cache = torch.cuda.ExternalStream(int(L.mi355_early_csr_stream()), device=self.device_)
        return so

    # ---------------------------------------------------------------------------------- table growth
    def _grow_target(self, incoming: int):
        """capacities the tables should have for their fill (as of the last completed read-back) + `incoming` keys, or None"""
        tb = self.table
        if getattr(self, "_fill_event", None) is None or not self._fill_event.query():
            return None
        sizes = self._fill_host.tolist()
        lf = self._dynamicemb_options[0].max_load_factor
        caps = list(tb.per_table_capacity_)
        new_caps = list(caps)
        per_table_in = incoming / max(self.num_tables, 1)
        for t in range(self.num_tables):
            while new_caps[t] < self._max_caps[t] and (sizes[t] + per_table_in) / new_caps[t] > lf:
                new_caps[t] = min(2 * new_caps[t], self._max_caps[t])
        return new_caps if new_caps != caps else None

    def _maybe_grow(self, incoming: int) -> None:
        """Grow the tables whose fill (as of the previous step: the sizes travel to pinned memory asynchronously, no sync on
        the step) plus the incoming keys passes max_load_factor: capacity doubles, by rehash, up to max_capacity.  A rehash
        moves every row, so it needs a point where no step holds slots / row addresses of the current table: with steps in
        flight it is DEFERRED to the first backward that leaves none (_grow_at_safe_point) -- under the prefetch pipeline that
        point still has the NEXT batch queued, whose prefetch is then dropped and re-resolved against the grown table."""
        tb = self.table
        new_caps = self._grow_target(incoming)
        if new_caps is not None:
            if not self._prefetch_states and len(self._live_steps) == 0:
                self._expand(new_caps)
                tb = self.table
                self._grow_deferred = False
                self._grow_skips = 0
            elif len(self._live_steps - set(self._prefetch_states)) == 0 and not getattr(self, "_regrowing", False):
                # only QUEUED prefetches hold the table (the usual state when the pipeline prefetches batch k + 1: batch k is
                # prefetched, not forwarded yet): grow NOW, before this batch's unseen keys are inserted into a table that is
                # too small for them (they would evict), and resolve the queued batches again
                self._grow_at_safe_point(new_caps)
                tb = self.table
            else:
                self._grow_skips = getattr(self, "_grow_skips", 0) + 1   # consecutive deferrals (reset by a successful expand)
                if self._grow_skips == 64:
                    import warnings

                    warnings.warn("DynamicEmb: table growth has been waiting for live / prefetched steps for 64 steps "
                                  "(rows evict at max capacity of the current size until a step boundary is free)")
                self._grow_deferred = True     # retried at the end of the next backward that leaves no forward alive
        if getattr(self, "_fill_host", None) is None:
            self._fill_host = torch.zeros(self.num_tables, dtype=torch.int64).pin_memory()
        self._fill_host.copy_(ext.segmented_sum_cuda(tb.bucket_sizes, tb.table_bucket_offsets_), non_blocking=True)
        self._fill_event = torch.cuda.Event()
        self._fill_event.record(current_torch_stream())

    def _grow_at_safe_point(self, new_caps=None) -> None:
        """A deferred growth, at the end of a backward that leaves no forward waiting for its backward.  Batches that are only
        PREFETCHED (queued, not yet forwarded) hold slots of the table that is about to be replaced: they are dropped and
        prefetched again against the grown table (their keys travel with the state), so the prefetch pipeline -- which always
        has the next batch queued at this point -- grows too (round-4 advisor finding: it never did)."""
        if new_caps is None:
            if getattr(self, "_fill_event", None) is not None:
                self._fill_event.synchronize()
            new_caps = self._grow_target(0)
        if new_caps is None:
            self._grow_deferred = False
            return
        queued = [(st.indices, st.offsets) for st in self._prefetch_states]
        if any(ind is None for ind, _ in queued):
            return                                   # (a state without its keys cannot be re-resolved: keep waiting)
        self.reset_prefetch()
        self._expand(new_caps)
        self._grow_deferred = False
        self._grow_skips = 0
        self._regrowing = True                       # (the re-resolved batches must not start another growth under this one)
        try:
            for ind, off in queued:
                self.prefetch(ind, off)
        finally:
            self._regrowing = False

    def _expand(self, new_caps) -> None:
        """rehash into a table of `new_caps` rows per table (key_value_table.py:559-666: export, re-insert with the stored
        scores, move the rows); the value buffers grow in place (VMM), rows move to their new slots"""
        from .scored_hashtable import ScoreArg

        old_tb = self.table
        C = old_tb.bucket_capacity_
        content = [list(self._export_table(t)) for t in range(self.num_tables)]   # (keys, full rows, scores) copies
        new_tb = LinearBucketTable([int(c) for c in new_caps], [ScoreSpec("score", self._policy)], key_type=torch.int64,
                                   bucket_capacity=C, device=self.device_)
        for t in range(self.num_tables):
            cap_t, vd = new_tb.per_table_capacity_[t], self.value_dims[t]
            self._vmm[t].extend(cap_t * vd)
            self.values[t] = self._vmm[t].data().view(cap_t, vd)
            for keys, rows, scores in content[t]:
                tids = torch.full_like(keys, t)
                idx = new_tb.insert(keys, tids, ScoreArg("score", scores.contiguous(), ext.ScorePolicy.ASSIGN))
                ok = idx >= 0
                self.values[t][idx[ok]] = rows[ok].to(self.embedding_dtype)
        self.table = new_tb
        self.table_ptrs = torch.tensor([v.data_ptr() for v in self.values], dtype=torch.int64, device=self.device_)
        self._fused_aux = None            # sized by the table
        self._plan_invalidate()
        self._fill_event = None

    # ---------------------------------------------------------------------------------- fused forward / backward
    def _fused_scores(self):
        """(find_policy, insert_policy, score_value, use_count) of mi355_demb_forward_fused"""
        P = ext.ScorePolicy
        s = self._score_strategy
        if s == DynamicEmbScoreStrategy.TIMESTAMP:
            return P.GLOBAL_TIMER, P.GLOBAL_TIMER, 0, 0
        if s in (DynamicEmbScoreStrategy.STEP, DynamicEmbScoreStrategy.CUSTOMIZED):
            return P.ASSIGN, P.ASSIGN, (self._step if s == DynamicEmbScoreStrategy.STEP else self._custom_score), 0
        if s == DynamicEmbScoreStrategy.LFU:
            return P.ACCUMULATE, P.ASSIGN, 0, 1
        return P.LRU_LFU, P.LRU_LFU, 0, 1

    # ---------------------------------------------------------------------------------- pre-bound training step
    # Round 5: module.forward() + autograd cost 0.179 ms per C2 step against 0.128 ms of GPU time -- the host
    # was the bound: ~60 ctypes arguments marshalled per forward, a step object with a dozen fields, size queries, torch.empty
    # calls.  In the steady-state configuration (HBM storage, fused index stage, no admission / growth / prefetch / pinning /
    # safe-check) the step now goes through a PLAN (csrc/pipeline.hip: mi355_demb_plan_*): everything constant is bound once,
    # a step is `torch.empty` for the output + one 13-argument C call, the backward one 19-argument C call.
    # Reference: DynamicEmbeddingFunction.forward / backward, batched_dynamicemb_function.py:1042-1300.
    def _plan_invalidate(self):
        pl = self.__dict__.get("_plan")
        if pl is not None:
            lib().mi355_demb_plan_destroy(pl)
        self._plan = None

    def __del__(self):
        try:
            self._plan_invalidate()
        except Exception:
            pass

    def _plan_key_now(self):
        """everything a plan froze that somebody may replace or mutate later: the table objects (_expand / load), the scratch and
        value-pointer tensors, and the hyper-parameters bound at creation (initializer seed and parameters, the optimizer state's
        initial value, the score policy) -- a change rebuilds the plan at the next step"""
        ia = self.initializer_args
        return (id(self.table), self._fused_aux.data_ptr() if self._fused_aux is not None else 0, self.table_ptrs.data_ptr(),
                self._seed, self.initial_accumulator_value, self._score_strategy, ia.mode, ia.mean, ia.std_dev, ia.lower, ia.upper,
                ia.value)

    def _plan_build(self):
        L = lib()
        tb = self.table
        if self._fused_aux is None:
            self._fused_aux = torch.zeros(L.mi355_demb_aux_numel(tb.capacity_, tb.num_buckets_), dtype=torch.int32, device=self.device_)
        fp, ip, _sval, use_cnt = self._fused_scores()
        mode, p = self._init_params()
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        combiner = -1 if not pooled else (0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1)
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims)
        self._plan = L.mi355_demb_plan_create(
            ptr(tb.table_storage_), ptr(tb.table_bucket_offsets_), tb.bucket_capacity_, tb.num_scores_,
            ptr(tb.bucket_sizes), ptr(tb._ref_counter), tb._ref_counter.numel(), ptr(self._fused_aux),
            self._fused_aux.numel(), tb.num_buckets_,
            ptr(self.table_ptrs), ptr(self.table_value_dims), ptr(self.table_emb_dims), dt(self.embedding_dtype),
            self.max_D, max(self.value_dims), ptr(self.feature_offsets), self.num_tables,
            int(fp), int(ip), int(use_cnt), 0,
            mode, c_f(p[0]), c_f(p[1]), c_f(p[2]), c_f(p[3]), c_u64(self._seed), c_f(self.initial_accumulator_value),
            combiner, ptr(self.D_offsets_t), self.total_D, dt(self.output_dtype), int(al), self._opt_kind,
            c_f(self.beta1), c_f(self.beta2), c_f(self.eps), c_f(self.weight_decay))
        if not self._plan:
            raise RuntimeError("mi355_demb_plan_create failed")
        # what the plan froze: anything else that changes rebuilds it (the table objects are replaced by _expand / load)
        self._plan_key = self._plan_key_now()
        self._plan_pooled = pooled
        self._plan_state = ctypes.c_int(-1)
        self._plan_state_ref = ctypes.byref(self._plan_state)
        self._plan_step_score = self._score_strategy == DynamicEmbScoreStrategy.STEP
        self._plan_custom_score = self._score_strategy == DynamicEmbScoreStrategy.CUSTOMIZED
        self._plan_al_g = al
        self._plan_fwd = L.mi355_demb_plan_forward
        self._plan_stage = L.mi355_demb_plan_stage
        self._plan_recency = self._score_strategy in (DynamicEmbScoreStrategy.STEP, DynamicEmbScoreStrategy.TIMESTAMP)
        self._plan_bwd = L.mi355_demb_plan_backward
        return self._plan

    def _plan_forward(self, indices, offsets):
        plan = self._plan
        if plan is None or self._plan_key != self._plan_key_now():
            self._plan_invalidate()
            plan = self._plan_build()
        n = indices.numel()
        num_bags = offsets.numel() - 1
        B = num_bags // self.feature_num
        buf, ring = None, None
        if not torch.cuda.is_current_stream_capturing():
            slot = self._bwd_ring_next % 4
            if not self._bwd_busy[slot]:
                object.__setattr__(self, "_bwd_ring_next", self._bwd_ring_next + 1)
                buf = self._step_ring[slot]
                self._bwd_busy[slot] = True
                ring = (self._bwd_busy, slot)
        if self._plan_pooled:
            out = torch.empty(B, self.total_D, dtype=self.output_dtype, device=self.device_)
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=self.device_)
        sval = self._step if self._plan_step_score else (self._custom_score if self._plan_custom_score else 0)
        s_ = _raw_stream(_cur_device())
        rc = 1
        if not self._pf_c_used:
            if buf is not None:
                rc = self._plan_fwd(plan, indices.data_ptr(), n, offsets.data_ptr(), num_bags, B, sval, ext.TIMER_OVERRIDE,
                                    out.data_ptr(), buf.data_ptr(), buf.numel(), self._plan_state_ref, s_)
            if rc == 1:      # no ring slot free (more steps outstanding than the ring holds), or the slot's buffer is too small
                need = lib().mi355_demb_plan_step_bytes(plan, n)
                if ring is not None:
                    buf = self._step_ring[ring[1]] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=self.device_)
                else:
                    buf = torch.empty(need, dtype=torch.uint8, device=self.device_)
                rc = self._plan_fwd(plan, indices.data_ptr(), n, offsets.data_ptr(), num_bags, B, sval, ext.TIMER_OVERRIDE,
                                    out.data_ptr(), buf.data_ptr(), buf.numel(), self._plan_state_ref, s_)
            score = None
        else:
            # this module prefetches on the partitioned path: a plain forward between prefetched ones takes part in the eviction
            # limit like them (its keys score >= `score`; its own evictions spare the steps in flight)
            score, timer = self._step_score(sval, exact=False)
            buf, rc = self._plan_stage_call(plan, 0, indices, n, offsets, num_bags, B, sval, ext.TIMER_OVERRIDE, out, buf, ring, s_)
        if rc != 0:
            if ring is not None:
                ring[0][ring[1]] = False
            check(rc, "demb_plan_forward")
        st = _PlanStep(self, buf, n, self.num_tables, ring, offsets, B, num_bags)
        if score is not None:
            self._inflight[st] = score
        tok = self._plan_state.value
        if tok <= -2:            # path (c): lazy reverse indices; -(2 + epoch) names the step's overflow notice (settle())
            st.lazy = True
            st.epoch = -2 - tok
            st.indices, st.sval = indices, sval
        st.prepared = 1 if n > 0 else 0
        object.__setattr__(self, "_step", self._step + 1)   # (nn.Module.__setattr__ runs its Parameter / Module checks per assignment)
        return out, st

    # ---- round 6: the prefetch pipeline on the partitioned index path (reference: BatchedDynamicEmbeddingTablesV2.prefetch,
    # batched_dynamicemb_tables.py:1090-1137, driven by PrefetchTrainPipelineSparseDist, train_pipeline.py:533-692).  prefetch(k+1)
    # runs the INDEX STAGE of batch k+1 (probe + partition kernel: table mutations, row addresses, the backward's CSR) on the
    # caller's side stream under the backward of batch k; forward(k+1) is the gather alone.  The reference pins the rows of a
    # prefetched batch with a ref-counter atomic per key so that a later prefetch cannot evict them; with recency scores (STEP,
    # TIMESTAMP) the same guarantee costs nothing: every key of a step in flight scores at least that step's score, and an
    # eviction takes only slots that score below the oldest of them (FusedArgs::protect).  Other score policies, tiny batches
    # and every non-default configuration keep the pinning prefetch.
    def _device_clock_now(self) -> int:
        """the device's score clock (100 MHz ticks) estimated on the host: read once with a sync, then extrapolated"""
        import time

        if self._clock0 is None:
            self._clock0 = (ext.device_timestamp(), time.monotonic_ns())
        d0, h0 = self._clock0
        return d0 + (time.monotonic_ns() - h0) // 10

    def _step_score(self, sval, exact: bool):
        """-> (score every key of the step is guaranteed to reach, timer override for the step's kernels).  STEP: the step number.
        TIMESTAMP: a staged step stamps its keys with the host's estimate of the device clock (exact); a plain forward reads the
        clock on the device when it runs -- not before the estimate taken now, minus a margin for the calibration (1 ms)."""
        if self._plan_step_score:
            return sval, ext.TIMER_OVERRIDE
        if ext.TIMER_OVERRIDE:
            return ext.TIMER_OVERRIDE, ext.TIMER_OVERRIDE
        now = self._device_clock_now()
        return (now, now) if exact else (now - 100_000, 0)

    def _plan_stage_call(self, plan, stage, indices, n, offsets, num_bags, B, sval, timer, out, buf, ring, s_, fork_from=None, slot=-1):
        protect = min(self._inflight.values(), default=0xFFFFFFFFFFFFFFFF)
        op = out.data_ptr() if out is not None else None
        rc = 1
        if buf is not None:
            rc = self._plan_stage(plan, stage, protect, indices.data_ptr(), n, offsets.data_ptr(), num_bags, B, sval, timer, op,
                                  buf.data_ptr(), buf.numel(), self._plan_state_ref, fork_from, slot, s_)
        if rc == 1:
            need = lib().mi355_demb_plan_step_bytes(plan, n)
            if ring is not None:
                buf = self._step_ring[ring[1]] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=self.device_)
            else:
                buf = torch.empty(need, dtype=torch.uint8, device=self.device_)
            rc = self._plan_stage(plan, stage, protect, indices.data_ptr(), n, offsets.data_ptr(), num_bags, B, sval, timer, op,
                                  buf.data_ptr(), buf.numel(), self._plan_state_ref, fork_from, slot, s_)
        return buf, rc

    def _plan_prefetch(self, indices, offsets, side=None):
        """index stage of a later batch on the current stream -- or, side given, on that stream behind what the current one
        holds, ordered by the library's own events --; None: not a batch of the partitioned path"""
        plan = self._plan
        if plan is None or self._plan_key != self._plan_key_now():
            self._plan_invalidate()
            plan = self._plan_build()
        if not self._plan_recency:
            return None
        n = indices.numel()
        num_bags = offsets.numel() - 1
        B = num_bags // self.feature_num
        if lib().mi355_demb_forward_fused_partitions(n, self.num_tables, self.table.num_buckets_) <= 0:
            return None
        buf, ring = None, None
        slot = self._bwd_ring_next % 4
        if not self._bwd_busy[slot]:
            object.__setattr__(self, "_bwd_ring_next", self._bwd_ring_next + 1)
            buf = self._step_ring[slot]
            self._bwd_busy[slot] = True
            ring = (self._bwd_busy, slot)
        sval = self._step if self._plan_step_score else 0
        score, timer = self._step_score(sval, exact=True)
        if side is not None and ring is None:
            return None                  # (more steps outstanding than the ring holds: the caller's stream-managed prefetch)
        self._pf_c_used = True
        cur = _raw_stream(_cur_device())
        if side is None:
            buf, rc = self._plan_stage_call(plan, 1, indices, n, offsets, num_bags, B, sval, timer, None, buf, ring, cur)
        else:
            if buf is None:              # first use of the slot: allocate here, on the caller's stream, not inside the staged call
                need = lib().mi355_demb_plan_step_bytes(plan, n)
                buf = self._step_ring[ring[1]] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=self.device_)
            buf, rc = self._plan_stage_call(plan, 1, indices, n, offsets, num_bags, B, sval, timer, None, buf, ring, side.cuda_stream,
                                            fork_from=cur, slot=ring[1])
        if rc != 0:
            if ring is not None:
                ring[0][ring[1]] = False
            if rc == 3:          # nothing was launched: not eligible after all (long bags, fp16 rows ...)
                return None
            check(rc, "demb_plan_stage")
        st = _PlanStep(self, buf, n, self.num_tables, ring, offsets, B, num_bags)
        tok = self._plan_state.value
        st.lazy = True
        st.epoch = -2 - tok if tok < -2 else 0
        st.indices, st.sval, st.timer = indices, sval, timer
        st.staged = True
        st.ev_slot = ring[1] if side is not None else -1
        st.prepared = 1 if n > 0 else 0
        self._inflight[st] = score
        object.__setattr__(self, "_step", self._step + 1)
        return st

    def _plan_gather_staged(self, st):
        """forward of a batch whose index stage ran in prefetch(): the gather alone, on the current stream"""
        if st.event is not None:
            current_torch_stream().wait_event(st.event)
        if st.ring is None and st.event is not None:
            st.buf.record_stream(current_torch_stream())      # (a buffer outside the module's ring: allocated under the prefetch stream)
        n = st.num_keys
        if self._plan_pooled:
            out = torch.empty(st.batch_size, self.total_D, dtype=self.output_dtype, device=self.device_)
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=self.device_)
        rc = self._plan_stage(self._plan, 2, 0xFFFFFFFFFFFFFFFF, st.indices.data_ptr(), n, st.offsets.data_ptr(), st.num_bags,
                              st.batch_size, st.sval, st.timer, out.data_ptr(), st.buf.data_ptr(), st.buf.numel(),
                              self._plan_state_ref, None, st.ev_slot, _raw_stream(_cur_device()))
        if rc != 0:
            check(rc, "demb_plan_stage(gather)")
        return out

    def _plan_backward(self, st, grads):
        if not grads.is_contiguous():
            grads = grads.contiguous()
        object.__setattr__(self, "_iter_num", self._iter_num + 1)
        buf = st.buf
        s_ = _raw_stream(_cur_device())
        # (the step's overflow notice is read inside the call: st.epoch > 0 unless somebody looked at the step already)
        rc = self._plan_bwd(self._plan, buf.data_ptr(), buf.numel(), st.num_keys, st.offsets.data_ptr(), st.num_bags, st.batch_size,
                            grads.data_ptr(), grads.stride(0), _DT_CODE[grads.dtype], int(self._plan_al_g and grads.stride(0) % 4 == 0),
                            self.learning_rate, self.beta1, self.beta2, self.eps, self.weight_decay, self._iter_num, 1, st.epoch, s_)
        if rc == 2:      # a flooded partition: regroup the step on the per-slot-counter path, then the backward (nothing is skipped)
            ep, st.epoch = st.epoch, 0
            self._rerun_step(st, ep)
            rc = self._plan_bwd(self._plan, buf.data_ptr(), buf.numel(), st.num_keys, st.offsets.data_ptr(), st.num_bags,
                                st.batch_size, grads.data_ptr(), grads.stride(0), _DT_CODE[grads.dtype],
                                int(self._plan_al_g and grads.stride(0) % 4 == 0), self.learning_rate, self.beta1, self.beta2,
                                self.eps, self.weight_decay, self._iter_num, 1, 0, s_)
        st.epoch = 0
        if self._pf_c_used:
            self._inflight.pop(st, None)
        if rc != 0:
            check(rc, "demb_plan_backward")
        st.prepared = 0
        r = st.ring
        if r is not None:
            r[0][r[1]] = False
            st.ring = None

    def _forward_fused(self, indices: torch.Tensor, offsets: torch.Tensor, train: bool, prefetch_only: bool):
        L = lib()
        n = indices.numel()
        num_bags = offsets.numel() - 1
        B = num_bags // self.feature_num
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        dev, T, tb = self.device_, self.num_tables, self.table
        if self._fused_aux is None:
            self._fused_aux = torch.zeros(L.mi355_demb_aux_numel(tb.capacity_, tb.num_buckets_), dtype=torch.int32, device=dev)
        fwd_b = L.mi355_demb_forward_fused_workspace_bytes(n, T)
        bwd_b = L.mi355_demb_backward_workspace_bytes(n, self.max_D) if (train and not prefetch_only) else 0
        need = _FusedStep.nbytes(n, T, fwd_b, bwd_b)
        capturing = torch.cuda.is_current_stream_capturing()
        buf, ring = None, None
        # the side stream works on module-owned memory only (a ring of step buffers): a step dropped without its backward
        # can then never hand memory that side kernels still write back to the allocator
        if train and not capturing:
            slot = self._bwd_ring_next % len(self._step_ring)
            if not self._bwd_busy[slot]:
                self._bwd_ring_next += 1
                buf = self._step_ring[slot]
                if buf is None or buf.numel() < need:
                    buf = self._step_ring[slot] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=dev)
                self._bwd_busy[slot] = True
                ring = (self._bwd_busy, slot)
        # side stream for the unique numbering + CSR: off by default -- at C2 the side kernels and the gather slow each
        # other down by more than the overlap saves (measured: 192 us with, 174 us without); worth it when a dense model
        # runs between this forward and its backward
        use_side = int(ring is not None and self._fused_side and not prefetch_only and torch.is_grad_enabled())
        if buf is None:
            buf = torch.empty(need, dtype=torch.uint8, device=dev)
        st = _FusedStep(self, buf, n, T, fwd_b, bwd_b)
        st.ring = ring
        st.offsets, st.batch_size, st.num_bags = offsets, B, num_bags
        st.pinned = bool((self._pin or prefetch_only) and train)
        if prefetch_only:
            out, combiner = None, -2
        elif pooled:
            out = torch.empty(B, self.total_D, dtype=self.output_dtype, device=dev)
            combiner = 0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=dev)
            combiner = -1
        fp, ip, sval, use_cnt = self._fused_scores()
        mode, p = self._init_params()
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims)
        tok = ctypes.c_int(-1)
        need_freq = use_cnt
        head = (
            ptr(tb.table_storage_), ptr(tb.table_bucket_offsets_), tb.bucket_capacity_, tb.num_scores_,
            ptr(tb.bucket_sizes), ptr(tb._ref_counter), tb._ref_counter.numel(), ptr(self._fused_aux),
            self._fused_aux.numel(), tb.num_buckets_,
            ptr(self.table_ptrs), ptr(self.table_value_dims), ptr(self.table_emb_dims), dt(self.embedding_dtype),
            self.max_D, max(self.value_dims),
            ptr(indices), n, ptr(offsets), num_bags, B, ptr(self.feature_offsets), T,
            int(train), int(fp), int(ip), c_u64(int(sval)), int(use_cnt), c_u64(ext.TIMER_OVERRIDE), int(st.pinned),
            mode, c_f(p[0]), c_f(p[1]), c_f(p[2]), c_f(p[3]), c_u64(self._seed), c_f(self.initial_accumulator_value),
            combiner, ptr(self.D_offsets_t), self.total_D, ptr(out), dt(out) if out is not None else 0, int(al),
            st.p("rev"), st.p("uoff"), st.p("tids"), st.p("slots"), st.p("row_addr"), st.p("freq") if need_freq else None,
            st.p("csr_cnt"), st.p("csr_rank"), st.p("bwd_ws") if bwd_b else None, bwd_b, use_side)
        tail = (st.p("fwd_ws"), fwd_b)
        check(L.mi355_demb_forward_fused(*head, ctypes.byref(tok), *tail, stream()), "demb_forward_fused")
        st.lazy = tok.value <= -2
        if tok.value < -2:       # path (c): the step's overflow notice is read before its CSR is used (_FusedStep.settle)
            st.epoch = -2 - tok.value
            st.rerun_args = (head, tail)
            st.indices = indices     # (the re-run reads the batch again)
        st.token = tok.value if tok.value >= 0 else -1
        st.prepared = (2 + tok.value) if tok.value >= 0 else (1 if bwd_b and n > 0 else 0)
        if train:
            self._step += 1
            if self._dynamicemb_options[0].safe_check_mode != DynamicEmbCheckMode.IGNORE:
                self._safe_check(st)
        return out, st

    def _rerun_step(self, st, epoch: int) -> None:
        """a partition's record list flooded in this step's forward (a key stream that defeats the hash: never seen with real keys):
        its index stage is redone on the per-slot-counter path over the step's own buffers (csrc/fused_fwd.hip:
        mi355_demb_forward_fused_rerun); the forward's output stands, the backward that follows updates every row"""
        self.overflow_reruns = getattr(self, "overflow_reruns", 0) + 1
        if getattr(st, "plan_step", False):
            buf = st.buf
            check(lib().mi355_demb_plan_rerun(self._plan, st.indices.data_ptr(), st.num_keys, st.offsets.data_ptr(), st.num_bags,
                                              st.batch_size, st.sval, ext.TIMER_OVERRIDE, buf.data_ptr(), buf.numel(), epoch,
                                              stream()), "demb_plan_rerun")
        else:
            head, tail = st.rerun_args
            check(lib().mi355_demb_forward_fused_rerun(*head, epoch, *tail, stream()), "demb_forward_fused_rerun")
        st.lazy = False          # the re-run left reverse indices and ranks eagerly

    def _backward_fused(self, st, grads: torch.Tensor):
        grads = grads.contiguous()
        if st.epoch:
            st.settle()
        self._iter_num += 1
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        combiner = -1 if not pooled else (0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1)
        dim = self.max_D
        L = lib()
        prepared = st.prepared
        if prepared and st.bwd_ws_bytes:
            ws_p, ws_b, ws = st.p("bwd_ws"), st.bwd_ws_bytes, None
        else:       # a prefetched step (no CSR yet) or a second backward of the same step: group here
            prepared = 0
            st.join()
            st.materialize()
            ws_b = L.mi355_demb_backward_workspace_bytes(st.num_keys, dim)
            ws = torch.empty(ws_b, dtype=torch.uint8, device=grads.device)
            ws_p = ptr(ws)
        if prepared >= 2 and st.token < 0:
            prepared = 1     # somebody (a property access) joined already
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims) and grads.stride(0) % 4 == 0
        tb = self.table
        check(L.mi355_demb_backward(
            st.p("rev"), st.num_keys, st.p("uoff"), self.num_tables, ptr(st.offsets), st.num_bags, st.batch_size,
            ptr(grads), grads.stride(0), dt(grads), ptr(self.D_offsets_t), dim, combiner, st.p("row_addr"),
            dt(self.embedding_dtype), self._opt_kind, c_f(self.learning_rate), c_f(self.beta1), c_f(self.beta2),
            c_f(self.eps), c_f(self.weight_decay), self._iter_num, -1, 1, int(al),
            ptr(tb._ref_counter), tb._ref_counter.numel(), st.p("slots"), st.p("tids"), ptr(tb.table_bucket_offsets_),
            tb.bucket_capacity_, int(bool(st.pinned)), st.p("csr_cnt"), st.p("csr_rank"), int(prepared), ws_p, ws_b,
            stream()), "demb_backward")
        st.token = -1
        st.prepared = 0
        st.release_ring()

    # ---------------------------------------------------------------------------------- hybrid tiers
    def _forward_hybrid(self, indices: torch.Tensor, offsets: torch.Tensor, train: bool, prefetch_only: bool = False):
        """HybridStorage forward (key_value_table.py:2107-2403, _prefetch path of batched_dynamicemb_function.py:298-556):
        find in the HBM tier, then the host tier; unseen keys enter the HBM tier, its evictions spill (key, score, row)
        to the host tier, keys the HBM tier cannot take go to the host tier directly.  Orchestrated from Python over the
        per-op C ABI (it needs the miss counts on the host, as the reference does); the result is one row address per
        unique key, after which gather and backward are the same launches as for HBM-only storage.

        prefetch_only (_prefetch_cache_path, batched_dynamicemb_function.py:298-556): the tier walk of a LATER batch, no
        gather.  Every row it resolved -- in either tier -- is pinned (ref counters of that tier's table) until the batch's
        backward, so the evictions of the batches prefetched after it cannot move it; and while prefetched batches are
        outstanding no key is PROMOTED from the host tier (a promotion moves a row, and an outstanding step may hold its
        host address): promotion resumes with the first forward that finds the queue empty.

        Admission (round 4: `admit_strategy` over the two tiers, batched_dynamicemb_function.py:559-696): a key missing in BOTH
        tiers adds its batch frequency to the admission counter; only the admitted ones take the insert walk above, the others are
        served for this step from scratch rows the initializer fills (not stored, no gradient), as in _forward_admission."""
        from .scored_hashtable import ScoreArg

        n = indices.numel()
        num_bags = offsets.numel() - 1
        B = num_bags // self.feature_num
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        dev, T = self.device_, self.num_tables
        eb = torch.empty((), dtype=self.embedding_dtype).element_size()
        st = _StepCtx()
        st.offsets, st.num_keys, st.batch_size, st.num_bags = offsets, n, B, num_bags
        st.tids = st.slots = None
        st.pinned, st.event = False, None
        pins = []      # prefetch_only: (tier table, slots, table ids) of every row this batch resolved
        if prefetch_only:
            out, combiner = None, -2
        elif pooled:
            out = torch.empty(B, self.total_D, dtype=self.output_dtype, device=dev)
            combiner = 0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=dev)
            combiner = -1
        rng = ext.get_table_range(offsets, self.feature_offsets)
        ukeys, st.rev, st.uoff, st.csr_cnt, st.csr_rank = ext.segmented_unique_csr(indices, rng, T)
        nu = int(st.uoff[-1].item())
        st.row_addr = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
        fwd_addr, scratch = st.row_addr, None      # (admission: the forward also reads scratch rows the backward must not touch)
        admission = train and self._admit_strategy is not None
        if nu > 0:
            uk = ukeys[:nu].contiguous()
            tids = ext.expand_table_ids_cuda(st.uoff, nu)
            fp, fs, ip, isc, need_freq = self._scores(nu)
            if need_freq:
                fs = isc = st.csr_cnt[:nu].to(torch.int64)
            find = ScoreArg("score", None if fs is None else fs[:nu], fp)
            ins = ScoreArg("score", None if isc is None else isc[:nu], ip)
            addr = st.row_addr[:nu]
            _, f0, s0 = self.table.lookup(uk, tids, find)
            hit0 = f0.nonzero().squeeze(1)
            if hit0.numel():
                addr[hit0] = ext.row_addresses(s0[hit0], tids[hit0], self.table_ptrs, self.table_value_dims, eb)
                if prefetch_only:
                    pins.append((self.table, s0[hit0].contiguous(), tids[hit0].contiguous()))
            miss = (~f0).nonzero().squeeze(1)
            if miss.numel():
                k1, t1 = uk[miss].contiguous(), tids[miss].contiguous()
                find1 = ScoreArg("score", None if find.value is None else find.value[miss].contiguous(), fp)
                _, f1, s1 = self.table_host.lookup(k1, t1, find1)
                hit1 = f1.nonzero().squeeze(1)
                if hit1.numel():
                    addr[miss[hit1]] = ext.row_addresses(s1[hit1], t1[hit1], self.table_ptrs_host, self.table_value_dims, eb)
                new = miss[(~f1).nonzero().squeeze(1)]
                rej = new[:0]
                if admission and new.numel():
                    km, tm = uk[new].contiguous(), tids[new].contiguous()
                    acc = self._admission_counter.add(km, tm, st.csr_cnt[:nu].to(torch.int64)[new].contiguous())
                    admit = self._admit_strategy.admit(km, acc)
                    rej, new = new[~admit], new[admit]
                    if new.numel():
                        self._admission_counter.erase(uk[new].contiguous(), tids[new].contiguous())
                # cache mode: host-tier hits are promoted into the HBM tier together with the unseen keys (never while a
                # prefetched batch is outstanding: it may hold the host address of the row a promotion would move)
                may_promote = train and self._promote and not prefetch_only and self._tier_prefetched == 0
                prom = miss[hit1] if may_promote else miss[:0]
                if prefetch_only and hit1.numel():
                    pins.append((self.table_host, s1[hit1].contiguous(), t1[hit1].contiguous()))
                cand = torch.cat([new, prom]) if prom.numel() else new
                if train and cand.numel():
                    kn, tn = uk[cand].contiguous(), tids[cand].contiguous()
                    n_new = new.numel()
                    insn = ScoreArg("score", None if ins.value is None else ins.value[cand].contiguous(), ip)
                    vmax = max(self.value_dims)
                    if prom.numel():   # their rows leave the host tier first: nothing below may overwrite them
                        buf_prom = torch.empty(prom.numel(), vmax, dtype=self.embedding_dtype, device=dev)
                        ext.load_from_flat_table_value(self.table_ptrs_host, s1[hit1].contiguous(), t1[hit1].contiguous(), buf_prom,
                                                       self.table_value_dims, self.table_emb_dims, self.max_D, True)
                    # rows found in the HBM tier for THIS batch must not be evicted by this batch's inserts
                    if hit0.numel():
                        self.table.increment_counter(s0[hit0].contiguous(), tids[hit0].contiguous())
                    idxn, h, ek, ei, es, et = self.table.insert_and_evict(kn, tn, insn)
                    if hit0.numel():
                        self.table.decrement_counter(s0[hit0].contiguous(), tids[hit0].contiguous())
                    buf_ev = None
                    if h:
                        ev = (ei >= 0).nonzero().squeeze(1)   # real evictions (negative entries mark refused inputs)
                        if ev.numel():   # take the evicted rows (embedding + optimizer state) out before they are overwritten
                            e_k, e_s, e_sc, e_t = ek[ev].contiguous(), ei[ev].contiguous(), es[ev].contiguous(), et[ev].contiguous()
                            buf_ev = torch.empty(ev.numel(), vmax, dtype=self.embedding_dtype, device=dev)
                            ext.load_from_flat_table_value(self.table_ptrs, e_s, e_t, buf_ev, self.table_value_dims,
                                                           self.table_emb_dims, self.max_D, True)
                    okn = (idxn >= 0).nonzero().squeeze(1)
                    if okn.numel():
                        addr[cand[okn]] = ext.row_addresses(idxn[okn].contiguous(), tn[okn].contiguous(), self.table_ptrs,
                                                            self.table_value_dims, eb)
                        if prefetch_only:
                            pins.append((self.table, idxn[okn].contiguous(), tn[okn].contiguous()))
                    if prom.numel():
                        pok = (idxn[n_new:] >= 0).nonzero().squeeze(1)   # promoted keys the HBM tier accepted
                        if pok.numel():
                            ext.store_to_flat_table_value(self.table_ptrs, idxn[n_new:][pok].contiguous(), tn[n_new:][pok].contiguous(),
                                                          buf_prom[pok].contiguous(), self.table_value_dims, self.table_emb_dims,
                                                          self.max_D, True)
                            self.table_host.erase(kn[n_new:][pok].contiguous(), tn[n_new:][pok].contiguous())
                        # the others stay where they are (their host address is already in `addr`)
                    if buf_ev is not None:   # spill the evicted rows into the host tier
                        dst = self.table_host.insert(e_k, e_t, ScoreArg("score", e_sc, ext.ScorePolicy.ASSIGN))
                        ok = (dst >= 0).nonzero().squeeze(1)
                        if ok.numel():
                            ext.store_to_flat_table_value(self.table_ptrs_host, dst[ok].contiguous(), e_t[ok].contiguous(),
                                                          buf_ev[ok].contiguous(), self.table_value_dims, self.table_emb_dims,
                                                          self.max_D, True)
                    bad = (idxn[:n_new] < 0).nonzero().squeeze(1)
                    if bad.numel():   # the HBM tier refused them (bucket full of pinned rows): they live in the host tier
                        insb = ScoreArg("score", None if insn.value is None else insn.value[:n_new][bad].contiguous(), ip)
                        sh = self.table_host.insert(kn[:n_new][bad].contiguous(), tn[:n_new][bad].contiguous(), insb)
                        okb = (sh >= 0).nonzero().squeeze(1)
                        if okb.numel():
                            addr[new[bad[okb]]] = ext.row_addresses(sh[okb].contiguous(), tn[:n_new][bad[okb]].contiguous(),
                                                                    self.table_ptrs_host, self.table_value_dims, eb)
                            if prefetch_only:
                                pins.append((self.table_host, sh[okb].contiguous(), tn[:n_new][bad[okb]].contiguous()))
                    if n_new:
                        mode, p = self._init_params()
                        a_new = addr[new].contiguous()
                        ext.init_rows(mode, p, self._seed, self.initial_accumulator_value, kn[:n_new].contiguous(), a_new,
                                      self.embedding_dtype, self.max_D, max(self.value_dims), skip=(a_new == 0),
                                      table_ids=tn[:n_new].contiguous(), table_emb_dims=self.table_emb_dims,
                                      table_value_dims=self.table_value_dims)
                if rej.numel():     # not admitted: scratch rows from the strategy's initializer (None: the table's) for this step only
                    vmax = max(self.value_dims)
                    scratch = torch.empty(rej.numel(), vmax, dtype=self.embedding_dtype, device=dev)
                    a_rej = scratch.data_ptr() + torch.arange(rej.numel(), dtype=torch.int64, device=dev) * (vmax * eb)
                    mode, p = self._init_params(getattr(self._admit_strategy, "initializer_args", None))
                    ext.init_rows(mode, p, self._seed, self.initial_accumulator_value, uk[rej].contiguous(), a_rej,
                                  self.embedding_dtype, self.max_D, vmax, table_ids=tids[rej].contiguous(),
                                  table_emb_dims=self.table_emb_dims, table_value_dims=self.table_value_dims)
                    fwd_addr = st.row_addr.clone()
                    fwd_addr[rej] = a_rej
        if prefetch_only:
            if admission:
                st.fwd_addr, st.scratch = fwd_addr, scratch
            self._pin_tier_rows(st, pins)
            self._step += 1
            return None, st
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims)
        if pooled:
            check(lib().mi355_gather_pooled(None, 0, ptr(fwd_addr), dt(self.embedding_dtype), ptr(st.rev), n, ptr(offsets),
                                            num_bags, B, combiner, self.max_D, ptr(self.D_offsets_t), self.total_D, ptr(out),
                                            dt(out), int(al), stream()), "gather_pooled")
        elif n:
            check(lib().mi355_gather_rows(None, 0, ptr(fwd_addr), dt(self.embedding_dtype), ptr(st.rev), n, None, self.max_D,
                                          ptr(out), out.stride(0), dt(out), int(al), stream()), "gather_rows")
        del scratch   # stream-ordered: the gather above is already queued
        if train:
            self._step += 1
        return out, st

    @staticmethod
    def _step_tensors(st):
        """every device tensor a step context holds (the arrays its gather / backward read)"""
        if isinstance(st, _FusedStep):
            return [st.buf] if isinstance(st.buf, torch.Tensor) else []
        out = []
        for name in _StepCtx.__slots__:
            v = getattr(st, name, None)
            if isinstance(v, torch.Tensor) and v.is_cuda:
                out.append(v)
        return out

    def _record_step_on(self, st, stream_) -> None:
        """A PREFETCHED step's arrays were allocated under the prefetch stream and are read by the gather / backward on
        another one: tell the caching allocator (record_stream), or a block freed with the step could be handed to the
        next prefetch while those kernels still read it."""
        if stream_ is None or getattr(st, "event", None) is None:
            return
        for t in self._step_tensors(st):
            t.record_stream(stream_)

    def _pin_tier_rows(self, st, pins) -> None:
        """rows a prefetched batch resolved through the Python-orchestrated paths (tiers, admission): pinned in their tier's
        table until the batch's backward (_release_tier_rows), like the `pinned` steps of the one-call pipeline.  A step that
        dies WITHOUT a backward (exception, eval() with batches queued, a drained pipeline) hands its pins to
        `_orphan_pins` through a finalizer; they are released at the module's next call (on a stream of ours, not in the
        garbage collector)."""
        import weakref

        for table, slots, tids in pins:
            table.increment_counter(slots, tids)
        st.tier_pins = pins
        self._tier_prefetched += 1
        cell = [pins]
        st.pin_cell = cell
        weakref.finalize(st, BatchedDynamicEmbeddingTablesV2._orphan_step, weakref.ref(self), cell)

    @staticmethod
    def _orphan_step(module_ref, cell) -> None:
        m = module_ref()
        if m is not None and cell[0] is not None:
            m._orphan_pins.append(cell[0])
            cell[0] = None

    def _drain_orphan_pins(self) -> None:
        while self._orphan_pins:
            for table, slots, tids in self._orphan_pins.pop():
                table.decrement_counter(slots, tids)
            self._tier_prefetched -= 1

    def _release_tier_rows(self, st) -> None:
        pins = getattr(st, "tier_pins", None)
        if pins is None:
            return
        st.tier_pins = None
        cell = getattr(st, "pin_cell", None)
        if cell is not None:
            cell[0] = None
        for table, slots, tids in pins:
            table.decrement_counter(slots, tids)
        self._tier_prefetched -= 1

    def reset_prefetch(self) -> None:
        """Drop every prefetched batch that has not been consumed: unpin its rows, release its ring slot.  Called when the
        module leaves training mode (train(False) / eval()) and usable by a pipeline that is drained early."""
        while self._prefetch_states:
            st = self._prefetch_states.popleft()
            self._live_steps.discard(st)
            self._release_tier_rows(st)
            if isinstance(st, _FusedStep) or getattr(st, "ring", None) is not None:
                st.release_ring()
        self._drain_orphan_pins()

    def train(self, mode: bool = True):
        if not mode and getattr(self, "_prefetch_states", None):
            self.reset_prefetch()
        return super().train(mode)

    def _forward_admission(self, indices: torch.Tensor, offsets: torch.Tensor, prefetch_only: bool = False):
        """Training forward with an admission strategy (_prefetch_hbm_direct_path, batched_dynamicemb_function.py:559-696,
        + DynamicEmbeddingFunction.forward :1090-1097): keys found in the table are served as usual; the batch frequency
        of every MISSING unique key is added to the admission counter, keys whose accumulated frequency passes
        `admit()` are erased from the counter and inserted (first-touch init), the others are NOT stored -- their
        embedding for this step is produced by the initializer into scratch rows and their gradients are dropped.
        Orchestrated from Python over the per-op C ABI (the admitted / rejected split is read on the host, as the
        reference does with its boolean-mask indexing); gather and backward are the launches of the plain path, because
        they take row ADDRESSES and a scratch row is as good an address as a table row.
        prefetch_only: the same walk for a later batch without the gather; found and admitted rows stay pinned until the
        batch's backward, the scratch rows of the rejected keys travel in the step context."""
        from .scored_hashtable import ScoreArg

        n = indices.numel()
        num_bags = offsets.numel() - 1
        B = num_bags // self.feature_num
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        dev, T = self.device_, self.num_tables
        eb = torch.empty((), dtype=self.embedding_dtype).element_size()
        st = _StepCtx()
        st.offsets, st.num_keys, st.batch_size, st.num_bags = offsets, n, B, num_bags
        st.tids = st.slots = None
        st.pinned, st.event = False, None
        pins = []
        if prefetch_only:
            out, combiner = None, -2
        elif pooled:
            out = torch.empty(B, self.total_D, dtype=self.output_dtype, device=dev)
            combiner = 0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=dev)
            combiner = -1
        rng = ext.get_table_range(offsets, self.feature_offsets)
        ukeys, st.rev, st.uoff, st.csr_cnt, st.csr_rank = ext.segmented_unique_csr(indices, rng, T)
        nu = int(st.uoff[-1].item())
        st.row_addr = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)   # what the backward updates: stored rows only
        fwd_addr = st.row_addr
        scratch = None
        if nu > 0:
            uk = ukeys[:nu].contiguous()
            tids = ext.expand_table_ids_cuda(st.uoff, nu)
            fp, fs, ip, isc, need_freq = self._scores(nu)
            freq = st.csr_cnt[:nu].to(torch.int64)       # occurrences of every unique key in this batch
            if need_freq:
                fs = isc = freq
            find = ScoreArg("score", None if fs is None else fs[:nu], fp)
            addr = st.row_addr[:nu]
            _, f0, s0 = self.table.lookup(uk, tids, find)
            hit = f0.nonzero().squeeze(1)
            if hit.numel():
                addr[hit] = ext.row_addresses(s0[hit], tids[hit], self.table_ptrs, self.table_value_dims, eb)
                if prefetch_only:
                    pins.append((self.table, s0[hit].contiguous(), tids[hit].contiguous()))
            miss = (~f0).nonzero().squeeze(1)
            if miss.numel():
                km, tm = uk[miss].contiguous(), tids[miss].contiguous()
                acc = self._admission_counter.add(km, tm, freq[miss].contiguous())
                admit = self._admit_strategy.admit(km, acc)
                adm, rej = miss[admit], miss[~admit]
                if adm.numel():
                    ka, ta = uk[adm].contiguous(), tids[adm].contiguous()
                    self._admission_counter.erase(ka, ta)
                    ins = ScoreArg("score", None if isc is None else isc[:nu][adm].contiguous(), ip)
                    if hit.numel():   # rows found for THIS batch must not be evicted by this batch's inserts
                        self.table.increment_counter(s0[hit].contiguous(), tids[hit].contiguous())
                    idx = self.table.insert(ka, ta, ins)
                    if hit.numel():
                        self.table.decrement_counter(s0[hit].contiguous(), tids[hit].contiguous())
                    ok = (idx >= 0).nonzero().squeeze(1)
                    if ok.numel():
                        a_new = ext.row_addresses(idx[ok].contiguous(), ta[ok].contiguous(), self.table_ptrs, self.table_value_dims, eb)
                        addr[adm[ok]] = a_new
                        if prefetch_only:
                            pins.append((self.table, idx[ok].contiguous(), ta[ok].contiguous()))
                        mode, p = self._init_params()
                        ext.init_rows(mode, p, self._seed, self.initial_accumulator_value, ka[ok].contiguous(), a_new,
                                      self.embedding_dtype, self.max_D, max(self.value_dims), table_ids=ta[ok].contiguous(),
                                      table_emb_dims=self.table_emb_dims, table_value_dims=self.table_value_dims)
                    bad = adm[(idx < 0).nonzero().squeeze(1)]
                    if bad.numel():       # admitted but the table refused them (bucket full of pinned rows): served like
                        rej = torch.cat([rej, bad])   # a rejected key for this step
                if rej.numel():
                    # scratch rows: embedding from the strategy's initializer (None: the table's), not stored, no update
                    vmax = max(self.value_dims)
                    scratch = torch.empty(rej.numel(), vmax, dtype=self.embedding_dtype, device=dev)
                    a_rej = scratch.data_ptr() + torch.arange(rej.numel(), dtype=torch.int64, device=dev) * (vmax * eb)
                    mode, p = self._init_params(getattr(self._admit_strategy, "initializer_args", None))
                    ext.init_rows(mode, p, self._seed, self.initial_accumulator_value, uk[rej].contiguous(), a_rej,
                                  self.embedding_dtype, self.max_D, vmax, table_ids=tids[rej].contiguous(),
                                  table_emb_dims=self.table_emb_dims, table_value_dims=self.table_value_dims)
                    fwd_addr = st.row_addr.clone()
                    fwd_addr[rej] = a_rej
        if prefetch_only:
            st.fwd_addr, st.scratch = fwd_addr, scratch      # the forward gathers from these (scratch rows stay alive with the step)
            self._pin_tier_rows(st, pins)
            self._step += 1
            return None, st
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims)
        if pooled:
            check(lib().mi355_gather_pooled(None, 0, ptr(fwd_addr), dt(self.embedding_dtype), ptr(st.rev), n, ptr(offsets),
                                            num_bags, B, combiner, self.max_D, ptr(self.D_offsets_t), self.total_D, ptr(out),
                                            dt(out), int(al), stream()), "gather_pooled")
        elif n:
            check(lib().mi355_gather_rows(None, 0, ptr(fwd_addr), dt(self.embedding_dtype), ptr(st.rev), n, None, self.max_D,
                                          ptr(out), out.stride(0), dt(out), int(al), stream()), "gather_rows")
        del scratch   # stream-ordered: the gather above is already queued
        self._step += 1
        return out, st

    # ---------------------------------------------------------------------------------- prefetch
    def prefetch(self, indices: torch.Tensor, offsets: torch.Tensor, forward_stream: Optional[torch.cuda.Stream] = None,
                 batch_size_per_feature_per_rank=None, frequency_counters=None) -> None:
        """BatchedDynamicEmbeddingTablesV2.prefetch (batched_dynamicemb_tables.py:1090-1137): run the index stage of a
        later batch (dedup, find, insert + first-touch init of unseen keys, pin) on the CURRENT stream, typically a side
        stream, while earlier batches still compute.  The rows it touches stay pinned (ref-counters) until the batch's
        backward releases them, so a later prefetch cannot evict them.  forward() consumes the states in FIFO order and
        only gathers."""
        if not self.training:
            return
        st = None
        if (self._plan_ok and not self._pin and not self._orphan_pins and indices.dtype is torch.int64
                and offsets.dtype is torch.int64 and indices.is_contiguous() and offsets.is_contiguous() and indices.is_cuda
                and os.environ.get("MI355_PREFETCH_C", "1") != "0"):
            st = self._plan_prefetch(indices, offsets)
        if st is None:
            _, st = self._forward_impl(indices, offsets, train=True, prefetch_only=True)
        st.event = torch.cuda.Event()
        st.event.record(current_torch_stream())
        st.indices = indices   # keeps the key tensor alive until the forward
        # the step's arrays were allocated under THIS (prefetch) stream; the gather and the backward read them on
        # `forward_stream` (reference signature: prefetch(..., forward_stream)): make the allocator aware now, and again on
        # whatever stream actually consumes them (_gather_prefetched / _backward_impl)
        if forward_stream is not None and forward_stream != current_torch_stream():
            self._record_step_on(st, forward_stream)
            if isinstance(indices, torch.Tensor) and indices.is_cuda:
                indices.record_stream(forward_stream)
        self._prefetch_states.append(st)

    def prefetch_async(self, indices: torch.Tensor, offsets: torch.Tensor) -> None:
        """prefetch() on the module's own prefetch stream, ordered behind what the CURRENT stream holds at the call (the batch,
        and the backward whose rows must be final before an eviction may recycle one) -- the fork, the end-of-stage mark and the
        wait of the later forward are library-owned events, no stream context is switched on the host.  Call it where the
        reference's pipeline calls prefetch: after forward(batch k) has been issued, before backward(batch k)."""
        if not self.training:
            return
        side = self.__dict__.get("_pf_stream")
        if side is None:
            side = torch.cuda.Stream(device=self.device_)
            object.__setattr__(self, "_pf_stream", side)
        st = None
        if (self._plan_ok and not self._pin and not self._orphan_pins and indices.dtype is torch.int64
                and offsets.dtype is torch.int64 and indices.is_contiguous() and offsets.is_contiguous() and indices.is_cuda
                and os.environ.get("MI355_PREFETCH_C", "1") != "0"):
            st = self._plan_prefetch(indices, offsets, side)
        if st is None:                  # not a batch of the partitioned path: the stream-managed form
            cur = current_torch_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self.prefetch(indices, offsets, forward_stream=cur)
            return
        st.event = None
        self._prefetch_states.append(st)

    def _gather_prefetched(self, st):
        if st.event is not None:
            current_torch_stream().wait_event(st.event)
            self._record_step_on(st, current_torch_stream())
        dev = self.device_
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims)
        n = st.num_keys
        src_addr = getattr(st, "fwd_addr", None)      # admission: rejected keys are served from scratch rows
        if src_addr is None:
            src_addr = st.row_addr
        if pooled:
            out = torch.empty(st.batch_size, self.total_D, dtype=self.output_dtype, device=dev)
            combiner = 0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1
            check(lib().mi355_gather_pooled(None, 0, ptr(src_addr), dt(self.embedding_dtype), ptr(st.rev), n, ptr(st.offsets),
                                            st.num_bags, st.batch_size, combiner, self.max_D, ptr(self.D_offsets_t), self.total_D,
                                            ptr(out), dt(out), int(al), stream()), "gather_pooled")
        else:
            out = torch.empty(n, self.dims[0], dtype=self.output_dtype, device=dev)
            if n:
                check(lib().mi355_gather_rows(None, 0, ptr(src_addr), dt(self.embedding_dtype), ptr(st.rev), n, None,
                                              self.max_D, ptr(out), out.stride(0), dt(out), int(al), stream()), "gather_rows")
        if st.event is not None and getattr(st, "scratch", None) is not None:
            for t in (st.scratch if isinstance(st.scratch, (list, tuple)) else [st.scratch]):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(current_torch_stream())   # allocated under the prefetch stream, read by the gather above
        st.scratch = None      # stream-ordered on THIS stream: the gather is queued
        return out

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
        if getattr(st, "plan_step", False) and st.prepared == 1 and self._plan is not None:
            return self._plan_backward(st, grads)
        self._live_steps.discard(st)
        self._record_step_on(st, current_torch_stream())   # (prefetched steps only: see _record_step_on)
        try:
            return self._backward_impl_inner(st, grads)
        finally:
            # a growth that had to wait for live / prefetched steps (under the prefetch pipeline one always exists when
            # _maybe_grow runs) is retried at the first safe point: here, when none is left
            if self._grow_deferred and len(self._live_steps - set(self._prefetch_states)) == 0:
                self._grow_at_safe_point()

    def _backward_impl_inner(self, st, grads: torch.Tensor):
        if isinstance(st, _FusedStep):
            return self._backward_fused(st, grads)
        grads = grads.contiguous()
        self._iter_num += 1
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        combiner = -1 if not pooled else (0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1)
        dim = self.max_D
        prepared = getattr(st, "bwd_ws", None) is not None
        ws = st.bwd_ws if prepared else torch.empty(lib().mi355_demb_backward_workspace_bytes(st.num_keys, dim),
                                                    dtype=torch.uint8, device=grads.device)
        al = all(d % 4 == 0 for d in self.dims) and all(v % 4 == 0 for v in self.value_dims) and grads.stride(0) % 4 == 0
        tb = self.table
        check(lib().mi355_demb_backward(
            ptr(st.rev), st.num_keys, ptr(st.uoff), self.num_tables, ptr(st.offsets), st.num_bags, st.batch_size,
            ptr(grads), grads.stride(0), dt(grads), ptr(self.D_offsets_t), dim, combiner, ptr(st.row_addr),
            dt(self.embedding_dtype), self._opt_kind, c_f(self.learning_rate), c_f(self.beta1), c_f(self.beta2),
            c_f(self.eps), c_f(self.weight_decay), self._iter_num, -1, 1, int(al),
            ptr(tb._ref_counter), tb._ref_counter.numel(), ptr(st.slots), ptr(st.tids), ptr(tb.table_bucket_offsets_),
            tb.bucket_capacity_, int(bool(getattr(st, "pinned", False)) and st.slots is not None), ptr(st.csr_cnt),
            ptr(st.csr_rank), (2 + st.token) if (prepared and getattr(st, "token", -1) >= 0) else 0, ptr(ws), ws.numel(),
            stream()), "demb_backward")
        if prepared:
            st.bwd_ws = None   # consumed: a second backward of the same step would have to regroup
            st.release_ring()
        self._release_tier_rows(st)

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, per_sample_weights=None,
                feature_requires_grad=None, batch_size_per_feature_per_rank=None, total_unique_indices=None):
        if per_sample_weights is not None:
            raise NotImplementedError("per_sample_weights is not supported (nor by the reference's kernels)")
        if self.training and torch.is_grad_enabled():
            return _LookupFunction.apply(self, indices, offsets, self._empty_tensor)
        out, _ = self._forward_impl(indices, offsets, train=False)
        return out

    # ---------------------------------------------------------------------------------- inspection
    # ---------------------------------------------------------------------------------- checkpoint wire format
    # Files of the reference (batched_dynamicemb_tables.py:73-92,1262-1409; key_value_table.py:1134-1290):
    #   {table}_emb_keys.rank_R.world_size_W        int64  [n]
    #   {table}_emb_values.rank_R.world_size_W      fp32   [n, dim]
    #   {table}_emb_scores.rank_R.world_size_W      int64  [n]   (LRU / timestamp scores are stored as AGE = now - score)
    #   {table}_emb_opt_values.rank_R.world_size_W  fp32   [n, ckpt_state_dim]   (optim=True)
    #   {table}_opt_args.json                       optimizer hyper-parameters + evict_strategy + dist_type (rank 0)
    # all raw little-endian, no headers; on load every rank reads every file and keeps the keys with key % W == rank.
    def _ckpt_state_dim(self, dim: int) -> int:
        return {1: 0, 2: 2 * dim, 3: dim, 4: 1}[self._opt_kind]

    def _opt_args(self):
        name = {1: "sgd", 2: "adam", 3: "exact_adagrad", 4: "exact_row_wise_adagrad"}[self._opt_kind]
        a = {"opt_type": name, "lr": self.learning_rate}
        if self._opt_kind == 2:
            a.update(iters=self._iter_num, beta1=self.beta1, beta2=self.beta2, eps=self.eps, weight_decay=self.weight_decay)
        elif self._opt_kind in (3, 4):
            a.update(eps=self.eps, initial_accumulator_value=self.initial_accumulator_value)
        return a

    def _is_lru(self) -> bool:
        return self._score_strategy == DynamicEmbScoreStrategy.TIMESTAMP

    @staticmethod
    def _rank_world(pg):
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group=pg), dist.get_world_size(group=pg)
        return 0, 1

    def _export_table(self, table_id: int, batch: int = 1 << 16, threshold: Optional[int] = None):
        """yields (keys i64[n], rows [n, value_dim], scores i64[n]) of one logical table over all storage tiers;
        `threshold`: only slots whose score is >= threshold (table_export_batch's filter, export_batch.cu:88-124)"""
        for tb, vals in self._tiers():
            C = tb.bucket_capacity_
            b0, b1 = int(tb.table_bucket_offsets_cpu_[table_id]), int(tb.table_bucket_offsets_cpu_[table_id + 1])
            lo, hi = b0 * C, b1 * C
            v = vals[table_id]
            for off in range(lo, hi, batch):
                n = min(batch, hi - off)
                cnt, keys, scores, idx = ext.table_export_batch(tb.table_storage_, C, n, off, torch.int64, threshold, lo,
                                                                tb.num_scores_, 0)
                c = int(cnt.item())
                if c == 0:
                    continue
                rows = v[idx[:c].to(v.device)].to(self.device_)
                yield keys[:c], rows, scores[:c]

    # ---- incremental dump (batched_dynamicemb_tables.py:1166-1180,1432-1482; key_value_table.py:1977-2036) ----------
    def get_score(self):
        """{table: current score}: the device timestamp for TIMESTAMP-scored tables (the reference value to pass to
        incremental_dump later), the stored step / custom score otherwise."""
        s = self._score_strategy
        if s == DynamicEmbScoreStrategy.TIMESTAMP or (isinstance(s, tuple) and DynamicEmbScoreStrategy.TIMESTAMP in s):
            now = int(ext.device_timestamp())
            return {n: now for n in self._table_names}
        val = self._step if s == DynamicEmbScoreStrategy.STEP else self._custom_score
        return {n: int(val) for n in self._table_names}

    def incremental_dump(self, named_thresholds=None, pg=None):
        """-> ({table: (keys i64[n], embeddings [n, dim])}, {table: score now}) with the rows whose score is >= the
        table's threshold: for TIMESTAMP tables the rows touched since `get_score()` returned that threshold, otherwise
        an absolute score cut.  Tensors are on the CPU for one rank; with a process group of more than one rank every
        rank gets the concatenation of all ranks' rows (on the device), as the reference's all-gather does."""
        import warnings

        import torch.distributed as dist

        ret, scores = {}, {}
        now = self.get_score()
        rank, world = self._rank_world(pg)
        multi = pg is not None and world > 1
        for name, thr in (named_thresholds or {}).items():
            if name not in self._table_names:
                warnings.warn(f"incremental_dump: table_name '{name}' is not in this module (available: "
                              f"{self._table_names}); skipping.", UserWarning, stacklevel=2)
                continue
            t = self._table_names.index(name)
            ks, vs = [], []
            for keys, rows, _ in self._export_table(t, threshold=int(thr)):
                ks.append(keys)
                vs.append(rows[:, : self.dims[t]].to(self.embedding_dtype))
            k = torch.cat(ks) if ks else torch.empty(0, dtype=torch.int64, device=self.device_)
            v = torch.cat(vs) if vs else torch.empty(0, self.dims[t], dtype=self.embedding_dtype, device=self.device_)
            if multi:
                cnt = torch.tensor([k.numel()], dtype=torch.int64, device=self.device_)
                cnts = [torch.zeros_like(cnt) for _ in range(world)]
                dist.all_gather(cnts, cnt, group=pg)
                sizes = [int(c.item()) for c in cnts]
                m = max(max(sizes), 1)
                kp = torch.zeros(m, dtype=torch.int64, device=self.device_)
                vp = torch.zeros(m, self.dims[t], dtype=v.dtype, device=self.device_)
                kp[: k.numel()], vp[: k.numel()] = k, v
                kg = [torch.empty_like(kp) for _ in range(world)]
                vg = [torch.empty_like(vp) for _ in range(world)]
                dist.all_gather(kg, kp, group=pg)
                dist.all_gather(vg, vp, group=pg)
                k = torch.cat([a[:n] for a, n in zip(kg, sizes)])
                v = torch.cat([a[:n] for a, n in zip(vg, sizes)])
            else:
                k, v = k.cpu(), v.cpu()
            ret[name] = (k, v)
            scores[name] = now[name]
        return ret, scores

    def dump(self, save_dir: str, optim: bool = False, counter: bool = False, table_names: Optional[List[str]] = None,
             pg=None) -> None:
        import json
        import os

        rank, world = self._rank_world(pg)
        names = table_names if table_names is not None else self._table_names
        os.makedirs(save_dir, exist_ok=True)
        now = ext.device_timestamp()
        for t, name in enumerate(self._table_names):
            if name not in set(names):
                continue
            D = self.dims[t]
            base = lambda item: os.path.join(save_dir, f"{name}_emb_{item}.rank_{rank}.world_size_{world}")  # noqa: E731
            if rank == 0:
                meta = self._opt_args()
                meta["evict_strategy"] = "EvictStrategy.KLru" if self._is_lru() else "EvictStrategy.KCustomized"
                meta["dist_type"] = self._dynamicemb_options[t].dist_type
                if self._score_strategy == DynamicEmbScoreStrategy.STEP:
                    meta["step_score"] = self._step
                with open(os.path.join(save_dir, f"{name}_opt_args.json"), "w") as f:
                    json.dump(meta, f)
            cs = self._ckpt_state_dim(D)
            with open(base("keys"), "wb") as fk, open(base("values"), "wb") as fv, open(base("scores"), "wb") as fs:
                fo = open(base("opt_values"), "wb") if optim else None
                for keys, rows, scores in self._export_table(t):
                    fk.write(keys.cpu().numpy().tobytes())
                    fv.write(rows[:, :D].float().contiguous().cpu().numpy().tobytes())
                    sc = (now - scores) if self._is_lru() else scores
                    fs.write(sc.cpu().numpy().tobytes())
                    if fo is not None and cs:
                        fo.write(rows[:, D:D + cs].float().contiguous().cpu().numpy().tobytes())
                if fo is not None:
                    fo.close()

    def load(self, save_dir: str, optim: bool = False, counter: bool = False, table_names: Optional[List[str]] = None,
             pg=None) -> None:
        import glob
        import json
        import os

        import numpy as np

        from .scored_hashtable import ScoreArg

        rank, world = self._rank_world(pg)
        names = table_names if table_names is not None else self._table_names
        now = ext.device_timestamp()
        dev = self.device_
        for t, name in enumerate(self._table_names):
            if name not in set(names):
                continue
            key_files = sorted(glob.glob(os.path.join(save_dir, f"{name}_emb_keys.rank_*.world_size_*")))
            if not key_files:
                continue
            meta_path = os.path.join(save_dir, f"{name}_opt_args.json")
            meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
            use_opt = optim
            if meta.get("opt_type") and meta["opt_type"] != self._opt_args()["opt_type"]:
                print(f"Optimizer type mismatch: {meta['opt_type']} != {self._opt_args()['opt_type']}. Will not load optimizer states.")
                use_opt = False
            if meta.get("dist_type", "roundrobin") != self._dynamicemb_options[t].dist_type:
                raise ValueError(f"Input dist_type mismatch: checkpoint was dumped with {meta.get('dist_type')!r}")
            if "step_score" in meta and self._score_strategy == DynamicEmbScoreStrategy.STEP:
                self._step = max(self._step, int(meta["step_score"]))
            if use_opt and self._opt_kind == 2 and "iters" in meta:
                self._iter_num = int(meta["iters"])
            D, V = self.dims[t], self.value_dims[t]
            cs = self._ckpt_state_dim(D)
            for kf in key_files:
                suffix = kf[kf.index("_emb_keys") + len("_emb_keys"):]
                path = lambda item: os.path.join(save_dir, f"{name}_emb_{item}{suffix}")  # noqa: E731
                nkeys = os.path.getsize(kf) // 8
                B = 1 << 16
                with open(kf, "rb") as fk, open(path("values"), "rb") as fv:
                    fs = open(path("scores"), "rb") if os.path.exists(path("scores")) else None
                    fo = open(path("opt_values"), "rb") if (use_opt and cs and os.path.exists(path("opt_values"))) else None
                    for start in range(0, nkeys, B):
                        n = min(B, nkeys - start)
                        keys = np.frombuffer(fk.read(8 * n), dtype=np.int64)
                        emb = np.frombuffer(fv.read(4 * D * n), dtype=np.float32).reshape(n, D)
                        sc = np.frombuffer(fs.read(8 * n), dtype=np.int64) if fs else None
                        op = np.frombuffer(fo.read(4 * cs * n), dtype=np.float32).reshape(n, cs) if fo else None
                        if world > 1:
                            m = (keys % world) == rank
                            keys, emb = keys[m], emb[m]
                            sc = sc[m] if sc is not None else None
                            op = op[m] if op is not None else None
                        if keys.size == 0:
                            continue
                        k_t = torch.from_numpy(keys.copy()).to(dev)
                        rows = torch.full((keys.size, V), float(self.initial_accumulator_value), dtype=torch.float32, device=dev)
                        rows[:, :D] = torch.from_numpy(emb.copy()).to(dev)
                        if op is not None:
                            rows[:, D:D + cs] = torch.from_numpy(op.copy()).to(dev)
                        elif self._opt_kind == 2:
                            rows[:, D:] = 0.0
                        if sc is not None:
                            s_t = torch.from_numpy(sc.copy()).to(dev)
                            s_t = (now - s_t) if self._is_lru() else s_t
                        else:
                            s_t = torch.full((keys.size,), now if self._is_lru() else 0, dtype=torch.int64, device=dev)
                        self._insert_rows(t, k_t, rows.to(self.embedding_dtype), s_t)
                    if fs:
                        fs.close()
                    if fo:
                        fo.close()

    def flush(self) -> None:
        """write-back of a promoting cache (batched_dynamicemb_tables.py:955); the tiers here hold each key once"""

    def export_keys_values(self, table_name: str, device: torch.device, batch_size: int = 65536):
        """(keys i64[n], embeddings fp32[n, dim]) of one table (batched_dynamicemb_tables.py:1411-1430)"""
        t = self._table_names.index(table_name)
        ks, vs = [], []
        for keys, rows, _ in self._export_table(t, batch_size):
            ks.append(keys.to(device))
            vs.append(rows[:, : self.dims[t]].float().to(device))
        if not ks:
            return torch.empty(0, dtype=torch.int64, device=device), torch.empty(0, self.dims[t], device=device)
        return torch.cat(ks), torch.cat(vs)

    def _insert_rows(self, table_id: int, keys: torch.Tensor, rows: torch.Tensor, scores: torch.Tensor) -> None:
        """insert (key, score) pairs and store their full rows: the first tier that takes a key keeps it"""
        from .scored_hashtable import ScoreArg

        tids = torch.full_like(keys, table_id)
        todo = torch.arange(keys.numel(), device=keys.device)
        for tb, vals in self._tiers():
            if todo.numel() == 0:
                break
            k, sc = keys[todo].contiguous(), scores[todo].contiguous()
            idx = tb.insert(k, tids[todo].contiguous(), ScoreArg("score", sc, ext.ScorePolicy.ASSIGN))
            ok = (idx >= 0).nonzero().squeeze(1)
            if ok.numel():
                v = vals[table_id]
                v[idx[ok].to(v.device)] = rows[todo[ok]].to(v.device)
            todo = todo[(idx < 0).nonzero().squeeze(1)]

    def _tiers(self):
        t = [(self.table, self.values)]
        if self.table_host is not None:
            t.append((self.table_host, self.values_host))
        return t

    def size(self, table_id: Optional[int] = None):
        return sum(tb.size(table_id).to(self.device_) for tb, _ in self._tiers())

    def lookup_rows(self, keys: torch.Tensor, table_id: int = 0):
        """(found, rows [n, value_dim]) of `keys` over all storage tiers -- test / debugging helper (CONST lookup)."""
        from .scored_hashtable import ScoreArg

        tids = torch.full_like(keys, table_id)
        found = torch.zeros(keys.numel(), dtype=torch.bool, device=keys.device)
        rows = torch.zeros(keys.numel(), self.value_dims[table_id], dtype=self.embedding_dtype, device=keys.device)
        for tb, vals in self._tiers():
            _, f, idx = tb.lookup(keys, tids, ScoreArg("score", None, ext.ScorePolicy.CONST))
            v = vals[table_id]
            r = v[idx.clamp(min=0).to(v.device)].to(keys.device)
            rows[f] = r[f]
            found |= f
        return found, rows
