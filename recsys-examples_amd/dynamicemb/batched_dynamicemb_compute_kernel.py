"""Compute-kernel wrappers of the TorchRec plugin surface: `BatchedDynamicEmbedding` (sequence) and
`BatchedDynamicEmbeddingBag` (pooled) present a group of dynamic tables of one rank as what TorchRec's grouped lookups hold
-- `forward(features: KJT) -> Tensor`, `fused_optimizer`, `named_parameters`, `state_dict`, `flush`, `purge`, `emb_module`
(reference batched_dynamicemb_compute_kernel.py:259-488) -- over `BatchedDynamicEmbeddingTablesV2`.

The grouped config is duck-typed: anything with `embedding_tables` (each: name, embedding_dim, local_rows, local_cols,
feature_names, fused_params with the table's `dynamicemb_options`), `data_type`, `pooling`, `fused_params` works --
TorchRec's GroupedEmbeddingConfig does, and so does `GroupedTables` below, which the sharded modules of
dynamicemb/shard build themselves."""
from __future__ import annotations

import warnings
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Any, Dict, Iterator, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn

from ._torchrec import DataType, EmptyFusedOptimizer, FusedOptimizer, PoolingType, data_type_to_dtype
from .batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2
from .dynamicemb_config import DynamicEmbPoolingMode, DynamicEmbTableOptions
from .planner import DynamicEmbParameterSharding


def pooling_mode_to_dynamicemb(pooling) -> DynamicEmbPoolingMode:
    """TorchRec PoolingType / FBGEMM PoolingMode (enum or its value) -> DynamicEmbPoolingMode"""
    name = getattr(pooling, "name", None)
    if name is None:
        name = {0: "SUM", 1: "MEAN", 2: "NONE"}.get(pooling, str(pooling).upper())
    try:
        return DynamicEmbPoolingMode[name]
    except KeyError:
        raise Exception(f"Invalid pooling type {pooling}")


@dataclass
class ShardedTable:
    """one table of a group, as this rank holds it"""
    name: str
    embedding_dim: int
    local_rows: int
    local_cols: int
    feature_names: List[str]
    fused_params: Dict[str, Any]
    pooling: Any = PoolingType.SUM
    num_embeddings: int = 0

    def num_features(self) -> int:
        return len(self.feature_names)


@dataclass
class GroupedTables:
    """tables that share one BatchedDynamicEmbeddingTablesV2 (equal grouped option keys), features table-major"""
    embedding_tables: List[ShardedTable]
    data_type: Any = DataType.FP32
    pooling: Any = PoolingType.SUM
    is_weighted: bool = False
    fused_params: Dict[str, Any] = field(default_factory=dict)

    def feature_names(self) -> List[str]:
        return [f for t in self.embedding_tables for f in t.feature_names]

    def feature_table_map(self) -> List[int]:
        return [i for i, t in enumerate(self.embedding_tables) for _ in t.feature_names]


class DynamicEmbeddingFusedOptimizer(FusedOptimizer):
    """what DMP's keyed optimizer sees of a table whose update runs inside its backward: one param group carrying the
    learning rate, forwarded to the module on every step / zero_grad (LR schedulers write param_groups[0]['lr'])"""

    def __init__(self, emb_module: BatchedDynamicEmbeddingTablesV2, lr: float) -> None:
        self._emb_module = [emb_module]   # in a list: not a submodule of whoever holds the optimizer
        super().__init__({}, {}, [{"params": [], "lr": lr}])

    def zero_grad(self, set_to_none: bool = False) -> None:
        self._emb_module[0].set_learning_rate(self.param_groups[0]["lr"])

    def step(self, closure: Any = None) -> None:
        self._emb_module[0].set_learning_rate(self.param_groups[0]["lr"])


def _prepare_fused_params(fused_params: Optional[Dict[str, Any]]) -> Dict[str, Any]:
    """TorchRec's fused_params -> keyword arguments of BatchedDynamicEmbeddingTablesV2: planner-only keys dropped,
    `output_dtype` (a TorchRec DataType or SparseType) as torch dtype, `betas` split"""
    fp = dict(fused_params or {})
    DynamicEmbParameterSharding.pop_additional_fused_params(fp)
    od = fp.get("output_dtype")
    if od is not None and not isinstance(od, torch.dtype):
        fp["output_dtype"] = od.as_dtype() if hasattr(od, "as_dtype") else data_type_to_dtype(od)
    if "betas" in fp:
        fp["beta1"], fp["beta2"] = fp.pop("betas")
    return fp


def _table_options(table, data_type) -> DynamicEmbTableOptions:
    """the per-rank options the planner left in the table's fused_params, checked against the shard TorchRec describes"""
    o = table.fused_params["dynamicemb_options"]
    want = data_type_to_dtype(data_type)
    if o.embedding_dtype is not None and o.embedding_dtype != want:
        warnings.warn(f"Table {table.name!r}: embedding_dtype {o.embedding_dtype} != grouped config data_type ({want}).",
                      UserWarning, stacklevel=2)
    if o.dim is not None and table.local_cols != o.dim:
        warnings.warn(f"Table {table.name!r}: local_cols={table.local_cols} != dynamicemb_options.dim={o.dim}.", UserWarning,
                      stacklevel=2)
    if o.max_capacity is not None and table.local_rows != o.max_capacity:
        warnings.warn(f"Table {table.name!r}: local_rows={table.local_rows} != max_capacity={o.max_capacity}.", UserWarning,
                      stacklevel=2)
    return o


class _BatchedDynamicKernel(nn.Module):
    """shared part of the two kernels"""

    def __init__(self, config, pg: Optional[dist.ProcessGroup], device: Optional[torch.device],
                 pooling_mode: DynamicEmbPoolingMode) -> None:
        super().__init__()
        self._config, self._pg, self._device = config, pg, device
        fused = _prepare_fused_params(getattr(config, "fused_params", None))
        tables = list(config.embedding_tables)
        options = [_table_options(t, config.data_type) for t in tables]
        fmap = [i for i, t in enumerate(tables) for _ in range(t.num_features())]
        self._emb_module = BatchedDynamicEmbeddingTablesV2(table_options=options, pooling_mode=pooling_mode,
                                                           feature_table_map=fmap, table_names=[t.name for t in tables],
                                                           device=device, **fused)
        self._optim = DynamicEmbeddingFusedOptimizer(self._emb_module, fused.get("learning_rate", 0.01))
        # one placeholder parameter per table: the rows live in the hash table and are updated in the backward, so what
        # DMP / the optimizer wrappers iterate over is a (1, 1) meta tensor marked as "optimizer in backward"
        self._param_per_table: Dict[str, nn.Parameter] = OrderedDict()
        for t in tables:
            p = nn.Parameter(torch.empty((1, 1), device=torch.device("meta"), dtype=self._emb_module.embedding_dtype))
            p._in_backward_optimizers = [EmptyFusedOptimizer()]
            self._param_per_table[t.name] = p

    @property
    def emb_module(self) -> BatchedDynamicEmbeddingTablesV2:
        return self._emb_module

    @property
    def config(self):
        return self._config

    @property
    def fused_optimizer(self) -> FusedOptimizer:
        return self._optim

    def named_split_embedding_weights(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        for name, p in self._param_per_table.items():
            yield (f"{prefix}.{name}.weight" if prefix else f"{name}.weight"), p

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True) -> Iterator[Tuple[str, nn.Parameter]]:
        yield from self.named_split_embedding_weights(prefix, recurse, remove_duplicate)

    def named_buffers(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        yield from ()   # fused parameters are not buffers either: the state lives in the tables

    def state_dict(self, destination: Optional[Dict[str, Any]] = None, prefix: str = "", keep_vars: bool = False):
        """placeholders only: dynamic tables are saved with DynamicEmbDump / module.dump (key / value / score files)"""
        if destination is None:
            destination = OrderedDict()
        for name, p in self._param_per_table.items():
            destination[f"{prefix}{name}.weight"] = p
        return destination

    def flush(self) -> None:
        self._emb_module.flush()

    def purge(self) -> None:
        reset = getattr(self._emb_module, "reset_cache_states", None)
        if reset is not None:
            reset()

    def forward(self, features) -> torch.Tensor:
        return self._emb_module(indices=features.values().long(), offsets=features.offsets().long(),
                                per_sample_weights=features.weights_or_none())


class BatchedDynamicEmbedding(_BatchedDynamicKernel):
    """sequence embeddings of a group of tables: forward(KJT) -> [num_keys, dim]"""

    def __init__(self, config, pg: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None) -> None:
        super().__init__(config, pg, device, DynamicEmbPoolingMode.NONE)


class BatchedDynamicEmbeddingBag(_BatchedDynamicKernel):
    """pooled embeddings of a group of tables: forward(KJT) -> [batch, sum of dims]"""

    def __init__(self, config, pg: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None) -> None:
        super().__init__(config, pg, device, pooling_mode_to_dynamicemb(config.pooling))
