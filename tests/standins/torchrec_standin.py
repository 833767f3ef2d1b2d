"""Minimal stand-ins for the TorchRec types the DynamicEmb plugin surface touches, used ONLY when `torchrec` cannot be
imported (it is not installed in this image; the GPU box may have it).  They implement the protocol -- names, fields,
call signatures -- that `dynamicemb.shard`, `dynamicemb.planner`, `dynamicemb.get_planner` and the compute-kernel wrappers
are written against, so that the plugin can be built, planned, sharded and run without the third-party package, and so that
the tests can drive sharder -> lookup -> output.  Nothing here is a port of TorchRec: it is the smallest object model with
the same public attribute names (torchrec release/V1.5.0: sparse/jagged_tensor.py, modules/embedding_configs.py,
modules/embedding_modules.py, distributed/types.py, distributed/embedding_types.py, distributed/planner/types.py).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Any, Dict, Generic, Iterator, List, Optional, Tuple, TypeVar

import torch
import torch.distributed as dist
from torch import nn


# ------------------------------------------------------------------------------------------- configs
class DataType(enum.Enum):
    FP32 = "FP32"
    FP16 = "FP16"
    BF16 = "BF16"


def data_type_to_dtype(data_type: DataType) -> torch.dtype:
    return {DataType.FP32: torch.float32, DataType.FP16: torch.float16, DataType.BF16: torch.bfloat16}[data_type]


class PoolingType(enum.Enum):
    SUM = "SUM"
    MEAN = "MEAN"
    NONE = "NONE"


@dataclass
class BaseEmbeddingConfig:
    num_embeddings: int
    embedding_dim: int
    name: str = ""
    data_type: DataType = DataType.FP32
    feature_names: List[str] = field(default_factory=list)

    def num_features(self) -> int:
        return len(self.feature_names)


@dataclass
class EmbeddingConfig(BaseEmbeddingConfig):
    pass


@dataclass
class EmbeddingBagConfig(BaseEmbeddingConfig):
    pooling: PoolingType = PoolingType.SUM


# ------------------------------------------------------------------------------------------- jagged tensors
class JaggedTensor:
    def __init__(self, values: torch.Tensor, lengths: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None,
                 weights: Optional[torch.Tensor] = None):
        self._values, self._lengths, self._offsets, self._weights = values, lengths, offsets, weights

    def values(self) -> torch.Tensor:
        return self._values

    def lengths(self) -> torch.Tensor:
        if self._lengths is None:
            self._lengths = self._offsets[1:] - self._offsets[:-1]
        return self._lengths

    def offsets(self) -> torch.Tensor:
        if self._offsets is None:
            z = torch.zeros(1, dtype=self._lengths.dtype, device=self._lengths.device)
            self._offsets = torch.cat([z, torch.cumsum(self._lengths, 0)])
        return self._offsets

    def weights_or_none(self):
        return self._weights


class KeyedJaggedTensor:
    """keys x batch jagged values, feature-major: lengths / offsets index slot f * B + b."""

    def __init__(self, keys: List[str], values: torch.Tensor, weights: Optional[torch.Tensor] = None,
                 lengths: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None, stride: Optional[int] = None):
        assert lengths is not None or offsets is not None
        self._keys, self._values, self._weights, self._lengths, self._offsets = list(keys), values, weights, lengths, offsets
        n = (lengths.numel() if lengths is not None else offsets.numel() - 1)
        self._stride = stride if stride is not None else (n // len(keys) if keys else 0)

    @staticmethod
    def from_lengths_sync(keys, values, lengths, weights=None):
        return KeyedJaggedTensor(keys, values, weights=weights, lengths=lengths)

    @staticmethod
    def from_offsets_sync(keys, values, offsets, weights=None):
        return KeyedJaggedTensor(keys, values, weights=weights, offsets=offsets)

    def keys(self) -> List[str]:
        return self._keys

    def values(self) -> torch.Tensor:
        return self._values

    def weights_or_none(self):
        return self._weights

    def stride(self) -> int:
        return self._stride

    def variable_stride_per_key(self) -> bool:
        return False

    def lengths(self) -> torch.Tensor:
        if self._lengths is None:
            self._lengths = self._offsets[1:] - self._offsets[:-1]
        return self._lengths

    def offsets(self) -> torch.Tensor:
        if self._offsets is None:
            z = torch.zeros(1, dtype=self._lengths.dtype, device=self._lengths.device)
            self._offsets = torch.cat([z, torch.cumsum(self._lengths, 0)])
        return self._offsets

    def to(self, device, non_blocking: bool = False):
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)  # noqa: E731
        return KeyedJaggedTensor(self._keys, mv(self._values), mv(self._weights), mv(self._lengths), mv(self._offsets), self._stride)

    def permute(self, order: List[int], order_tensor: Optional[torch.Tensor] = None) -> "KeyedJaggedTensor":
        B = self._stride
        off = self.offsets().tolist()
        lens = self.lengths().view(len(self._keys), B)
        vals = [self._values[off[f * B]: off[(f + 1) * B]] for f in order]
        w = None if self._weights is None else torch.cat([self._weights[off[f * B]: off[(f + 1) * B]] for f in order])
        return KeyedJaggedTensor([self._keys[f] for f in order], torch.cat(vals) if vals else self._values[:0], w,
                                 lengths=lens[order].reshape(-1), stride=B)

    def split(self, segments: List[int]) -> List["KeyedJaggedTensor"]:
        out, f0, B = [], 0, self._stride
        off = self.offsets()
        for s in segments:
            lo, hi = int(off[f0 * B]), int(off[(f0 + s) * B])
            w = None if self._weights is None else self._weights[lo:hi]
            out.append(KeyedJaggedTensor(self._keys[f0:f0 + s], self._values[lo:hi], w,
                                         lengths=self.lengths()[f0 * B:(f0 + s) * B], stride=B))
            f0 += s
        return out

    def to_dict(self) -> Dict[str, JaggedTensor]:
        B, off = self._stride, self.offsets()
        d = {}
        for f, k in enumerate(self._keys):
            lo, hi = int(off[f * B]), int(off[(f + 1) * B])
            d[k] = JaggedTensor(self._values[lo:hi], lengths=self.lengths()[f * B:(f + 1) * B])
        return d


class KeyedTensor:
    """pooled embeddings [B, sum(length_per_key)] with one column block per key"""

    def __init__(self, keys: List[str], length_per_key: List[int], values: torch.Tensor, key_dim: int = 1):
        self._keys, self._lpk, self._values = list(keys), list(length_per_key), values

    def keys(self):
        return self._keys

    def length_per_key(self):
        return self._lpk

    def values(self):
        return self._values

    def to_dict(self) -> Dict[str, torch.Tensor]:
        d, o = {}, 0
        for k, n in zip(self._keys, self._lpk):
            d[k] = self._values[:, o:o + n]
            o += n
        return d


# ------------------------------------------------------------------------------------------- unsharded modules
class EmbeddingCollection(nn.Module):
    """the unsharded module a sharder replaces; on the `meta` device it is only a bag of configs"""

    def __init__(self, tables: List[EmbeddingConfig], device: Optional[torch.device] = None, need_indices: bool = False):
        super().__init__()
        self._embedding_configs = list(tables)
        self._device = device
        self._embedding_dim = tables[0].embedding_dim if tables else 0

    def embedding_configs(self) -> List[EmbeddingConfig]:
        return self._embedding_configs

    def embedding_dim(self) -> int:
        return self._embedding_dim


class EmbeddingBagCollection(nn.Module):
    def __init__(self, tables: List[EmbeddingBagConfig], is_weighted: bool = False, device: Optional[torch.device] = None):
        super().__init__()
        self._embedding_bag_configs = list(tables)
        self._device = device

    def embedding_bag_configs(self) -> List[EmbeddingBagConfig]:
        return self._embedding_bag_configs

    def embedding_configs(self) -> List[EmbeddingBagConfig]:   # (examples/commons/distributed/sharding.py:194 calls this on both)
        return self._embedding_bag_configs


# ------------------------------------------------------------------------------------------- distributed types
class ShardingType(enum.Enum):
    DATA_PARALLEL = "data_parallel"
    TABLE_WISE = "table_wise"
    COLUMN_WISE = "column_wise"
    ROW_WISE = "row_wise"
    TABLE_ROW_WISE = "table_row_wise"
    TABLE_COLUMN_WISE = "table_column_wise"


class EmbeddingComputeKernel(enum.Enum):
    DENSE = "dense"
    FUSED = "fused"
    FUSED_UVM = "fused_uvm"
    FUSED_UVM_CACHING = "fused_uvm_caching"
    QUANT = "quant"
    CUSTOMIZED_KERNEL = "customized_kernel"


class BoundsCheckMode(enum.IntEnum):
    FATAL = 0
    WARNING = 1
    IGNORE = 2
    NONE = 3


@dataclass
class ParameterConstraints:
    sharding_types: Optional[List[str]] = None
    compute_kernels: Optional[List[str]] = None
    min_partition: Optional[int] = None
    pooling_factors: List[float] = field(default_factory=lambda: [1.0])
    num_poolings: Optional[List[float]] = None
    batch_sizes: Optional[List[int]] = None
    is_weighted: bool = False
    cache_params: Any = None
    enforce_hbm: Optional[bool] = None
    stochastic_rounding: Optional[bool] = None
    bounds_check_mode: Optional[BoundsCheckMode] = None
    feature_names: Optional[List[str]] = None
    output_dtype: Any = None
    device_group: Optional[str] = None
    key_value_params: Any = None


@dataclass
class ShardMetadata:
    shard_offsets: List[int]
    shard_sizes: List[int]
    placement: Any = None


@dataclass
class EnumerableShardingSpec:
    shards: List[ShardMetadata]


def placement(compute_device: str, rank: int, local_size: int) -> str:
    return f"rank:{rank}/{compute_device}:{rank % max(local_size, 1)}" if compute_device == "cuda" else f"rank:{rank}/{compute_device}"


@dataclass
class ParameterSharding:
    sharding_type: str
    compute_kernel: str
    ranks: Optional[List[int]] = None
    sharding_spec: Optional[EnumerableShardingSpec] = None
    cache_params: Any = None
    enforce_hbm: Optional[bool] = None
    stochastic_rounding: Optional[bool] = None
    bounds_check_mode: Optional[BoundsCheckMode] = None
    output_dtype: Any = None
    key_value_params: Any = None


class EmbeddingModuleShardingPlan(dict):
    """table name -> ParameterSharding"""


@dataclass
class ShardingPlan:
    plan: Dict[str, EmbeddingModuleShardingPlan]

    def get_plan_for_module(self, module_path: str):
        return self.plan.get(module_path)


@dataclass
class Topology:
    world_size: int
    compute_device: str
    hbm_cap: Optional[int] = None
    ddr_cap: Optional[int] = None
    local_world_size: Optional[int] = None
    intra_host_bw: float = 0.0
    inter_host_bw: float = 0.0


@dataclass
class HeuristicalStorageReservation:
    percentage: float = 0.15


def get_local_size(world_size: Optional[int] = None) -> int:
    import os

    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    return int(os.environ.get("LOCAL_WORLD_SIZE", world_size))


class ShardingEnv:
    def __init__(self, world_size: int, rank: int, pg: Optional[dist.ProcessGroup] = None):
        self.world_size, self.rank, self.process_group = world_size, rank, pg

    @classmethod
    def from_process_group(cls, pg: dist.ProcessGroup) -> "ShardingEnv":
        return cls(dist.get_world_size(pg), dist.get_rank(pg), pg)


# ------------------------------------------------------------------------------------------- awaitables, optimizers
W = TypeVar("W")


class Awaitable(Generic[W]):
    def wait(self) -> W:
        return self._wait_impl()

    def _wait_impl(self) -> W:
        raise NotImplementedError


class LazyAwaitable(Awaitable[W]):
    pass


class NoWait(Awaitable[W]):
    def __init__(self, obj: W):
        self._obj = obj

    def _wait_impl(self) -> W:
        return self._obj


class FusedOptimizer(torch.optim.Optimizer):
    """optimizer whose step happens inside the backward of the module that owns it"""

    def __init__(self, params, state, param_groups):
        self._params, self.state, self.param_groups = params, state, param_groups
        self.defaults = {}

    def step(self, closure: Any = None) -> None:
        pass

    def zero_grad(self, set_to_none: bool = False) -> None:
        pass


class EmptyFusedOptimizer(FusedOptimizer):
    def __init__(self) -> None:
        super().__init__({}, {}, [])


class FusedOptimizerModule:
    @property
    def fused_optimizer(self):
        raise NotImplementedError


M = TypeVar("M", bound=nn.Module)


class ModuleSharder(Generic[M]):
    def __init__(self, qcomm_codecs_registry: Optional[Dict[str, Any]] = None) -> None:
        self._qcomm_codecs_registry = qcomm_codecs_registry

    @property
    def qcomm_codecs_registry(self):
        return self._qcomm_codecs_registry

    @property
    def module_type(self):
        raise NotImplementedError

    def shard(self, module, params, env, device=None, module_fqn=None):
        raise NotImplementedError

    def sharding_types(self, compute_device_type: str) -> List[str]:
        return [ShardingType.ROW_WISE.value, ShardingType.DATA_PARALLEL.value]

    def compute_kernels(self, sharding_type: str, compute_device_type: str) -> List[str]:
        return [k.value for k in EmbeddingComputeKernel]

    def shardable_parameters(self, module) -> Dict[str, nn.Parameter]:
        return {}


class _BaseCollectionSharder(ModuleSharder[M]):
    def __init__(self, fused_params: Optional[Dict[str, Any]] = None, qcomm_codecs_registry: Optional[Dict[str, Any]] = None,
                 **kwargs) -> None:
        super().__init__(qcomm_codecs_registry)
        self._fused_params = fused_params

    @property
    def fused_params(self):
        return self._fused_params


class EmbeddingCollectionSharder(_BaseCollectionSharder[EmbeddingCollection]):
    def __init__(self, fused_params=None, qcomm_codecs_registry=None, use_index_dedup: bool = False, **kwargs) -> None:
        super().__init__(fused_params, qcomm_codecs_registry)
        self._use_index_dedup = use_index_dedup

    @property
    def module_type(self):
        return EmbeddingCollection


class EmbeddingBagCollectionSharder(_BaseCollectionSharder[EmbeddingBagCollection]):
    @property
    def module_type(self):
        return EmbeddingBagCollection


class ShardedModule(nn.Module, FusedOptimizerModule):
    """input_dist -> compute -> output_dist, as torchrec.distributed.types.ShardedModule splits a forward"""

    def create_context(self):
        raise NotImplementedError

    def input_dist(self, ctx, *input, **kwargs):
        raise NotImplementedError

    def compute(self, ctx, dist_input):
        raise NotImplementedError

    def output_dist(self, ctx, output):
        raise NotImplementedError

    def compute_and_output_dist(self, ctx, input):
        return self.output_dist(ctx, self.compute(ctx, input))

    def forward(self, *input, **kwargs):
        ctx = self.create_context()
        dist_input = self.input_dist(ctx, *input, **kwargs).wait().wait()
        return self.compute_and_output_dist(ctx, dist_input)


class CombinedOptimizer:
    def __init__(self, optims: List[Tuple[str, Any]]):
        self._optims = optims

    @property
    def optimizers(self):
        return self._optims

    def step(self, closure: Any = None) -> None:
        for _, o in self._optims:
            o.step()

    def zero_grad(self, set_to_none: bool = False) -> None:
        for _, o in self._optims:
            o.zero_grad(set_to_none)

    @property
    def param_groups(self):
        return [g for _, o in self._optims for g in o.param_groups]


class DistributedModelParallel(nn.Module):
    """replaces every module a sharder knows (and the plan covers) by `sharder.shard(...)`; the rest stays as it is"""

    def __init__(self, module: nn.Module, env: Optional[ShardingEnv] = None, device: Optional[torch.device] = None,
                 plan: Optional[ShardingPlan] = None, sharders: Optional[List[ModuleSharder]] = None,
                 init_data_parallel: bool = True, init_parameters: bool = True, data_parallel_wrapper: Any = None) -> None:
        super().__init__()
        self._env = env if env is not None else ShardingEnv.from_process_group(dist.group.WORLD)
        self._plan = plan
        by_type = {s.module_type: s for s in (sharders or [])}
        self._dmp_wrapped_module = self._shard(module, "", by_type, device)

    def _shard(self, module: nn.Module, path: str, by_type, device) -> nn.Module:
        sh = by_type.get(type(module))
        mplan = self._plan.get_plan_for_module(path) if self._plan is not None else None
        if sh is not None and mplan is not None:
            return sh.shard(module, mplan, self._env, device, path)
        for name, child in list(module.named_children()):
            new = self._shard(child, f"{path}.{name}" if path else name, by_type, device)
            if new is not child:
                setattr(module, name, new)
        return module

    @property
    def module(self) -> nn.Module:
        return self._dmp_wrapped_module

    def forward(self, *args, **kwargs):
        return self._dmp_wrapped_module(*args, **kwargs)

    @property
    def fused_optimizer(self) -> CombinedOptimizer:
        return CombinedOptimizer([(n, m.fused_optimizer) for n, m in self._dmp_wrapped_module.named_modules()
                                  if isinstance(m, ShardedModule)])

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True) -> Iterator[Tuple[str, nn.Parameter]]:
        """a sharded module answers for its whole subtree (its tables are not nn.Parameters), everything else as usual"""
        def walk(m: nn.Module, pre: str):
            if isinstance(m, ShardedModule):
                yield from m.named_parameters(pre, recurse)
                return
            for n, p_ in m._parameters.items():
                if p_ is not None:
                    yield (f"{pre}.{n}" if pre else n), p_
            if recurse:
                for n, c in m.named_children():
                    yield from walk(c, f"{pre}.{n}" if pre else n)

        yield from walk(self._dmp_wrapped_module, prefix)


def in_backward_optimizer_filter(named_parameters, include: bool = False):
    for n, p in named_parameters:
        if hasattr(p, "_in_backward_optimizers") == include:
            yield n, p
