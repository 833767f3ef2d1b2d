"""Tables backed by a user-supplied key-value store (`DynamicEmbTableOptions.external_storage`).

Reference: the HOST_PS layout of BatchedDynamicEmbeddingTablesV2 (batched_dynamicemb_tables.py:698-721) and its generic
forward / backward (batched_dynamicemb_function.py:934-1040, 1190-1290): the module de-duplicates the batch, asks the store
for the rows of the unique keys, initialises and inserts the ones it does not have, pools from the dense value buffer, and in
the backward reduces the gradients per unique key, applies the optimizer to the buffer and writes the rows back.
Everything but `Storage.find` / `Storage.insert` (user code) runs in the HIP kernels of the per-op chain: segmented unique,
row initialisation, pooled / row gather, gradient reduction, the padded-buffer optimizers.
"""
import json
import os
from copy import deepcopy
from typing import List, Optional

import torch
from torch import nn

import dynamicemb_extensions as ext

from .batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2, init_dense_rows
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
    prefetch, table growth, admission and the HBM cache in front of the store (`caching=True`) are not available."""

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
        if opt0.caching:
            raise NotImplementedError("an HBM cache in front of an external storage (caching=True) is not supported")
        if opt0.admit_strategy is not None:
            raise NotImplementedError("admission with an external storage")
        self._dynamicemb_options = table_options
        self._table_names = table_names or [f"t{i}" for i in range(len(table_options))]
        self.pooling_mode, self.output_dtype, self.use_index_dedup = pooling_mode, output_dtype, use_index_dedup
        self.index_type = opt0.index_type or torch.int64
        self.embedding_dtype = opt0.embedding_dtype or torch.float32
        self.device_ = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dims: List[int] = [o.dim for o in table_options]
        if len(set(self.dims)) != 1:
            raise NotImplementedError("external storage with tables of different embedding dims")
        T_ = len(table_options)
        self.feature_table_map = feature_table_map if feature_table_map is not None else list(range(T_))
        assert all(any(t == m for m in self.feature_table_map) for t in range(T_)), "Each table must have at least one feature!"
        self.total_D = sum(self.dims[t] for t in self.feature_table_map)
        self.max_D, self.mixed_D = self.dims[0], False
        self.D_offsets_t = None
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
        n, num_bags = indices.numel(), offsets.numel() - 1
        B = num_bags // self.feature_num
        T, D, V, dev = self.num_tables, self.dims[0], self.value_dims[0], self.device_
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
                mode, p = self._init_params()
                addr = (values.data_ptr() + midx * (values.stride(0) * values.element_size())).contiguous()
                ext.init_rows(mode, p, self._seed, float(self.initial_accumulator_value), mkeys.to(torch.int64).contiguous(), addr,
                              values.dtype, D, values.size(1))
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
                ext.gather_embedding_pooled(emb, out, rev, offsets, combiner, self.total_D, B, None, D)
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

    def _backward_impl(self, st, grads: torch.Tensor):
        if st is None or st.nu == 0:
            return
        D, V = self.dims[0], self.value_dims[0]
        pooled = self.pooling_mode != DynamicEmbPoolingMode.NONE
        combiner = (0 if self.pooling_mode == DynamicEmbPoolingMode.SUM else 1) if pooled else -1
        g = ext.reduce_grads(st.rev, grads.contiguous(), st.nu, st.batch_size, D, st.offsets if pooled else None, None, combiner,
                             self.total_D)
        vals, tid = st.values, st.tids
        dims_t = torch.tensor(self.dims, dtype=torch.int64, device=self.device_)
        al = D % 4 == 0 and V % 4 == 0
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

    # ------------------------------------------------------------------ the rest of the surface
    def prefetch(self, *args, **kwargs):
        raise NotImplementedError("prefetch with an external storage")

    def reset_prefetch(self) -> None:
        return

    def train(self, mode: bool = True):
        return nn.Module.train(self, mode)

    def flush(self) -> None:
        return

    def size(self, table_id: Optional[int] = None):
        return self._storage.size()

    @property
    def storage(self):
        return self._storage

    def _paths(self, save_dir: str, name: str, pg):
        rank, world = self._rank_world(pg)
        base = lambda item: os.path.join(save_dir, f"{name}_emb_{item}.rank_{rank}.world_size_{world}")  # noqa: E731
        return os.path.join(save_dir, f"{name}_opt_args.json"), base("keys"), base("values"), base("scores"), base("opt_values")

    def dump(self, save_dir: str, optim: bool = False, counter: bool = False, table_names=None, pg=None) -> None:
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

    def export_keys_values(self, table_name: str, device: torch.device, batch_size: int = 65536):
        return self._storage.export_keys_values(device, batch_size, self._table_names.index(table_name))
