"""DynamicEmbeddingShardingPlanner and its constraint / plan types (reference planner/planner.py:63-387).

A dynamic embedding table has exactly one admissible plan: ROW_WISE over every rank of the group (the only model-parallel
mode of the reference, README.md:99), computed by the DynamicEmb kernel, each rank owning ceil(N / W) rows rounded up to
whole hash buckets and global_hbm_for_values / W bytes of HBM for values.  So nothing is searched for them: this planner
fills the per-rank table options in, emits that plan entry, and leaves every other table to TorchRec's planner (or, when
TorchRec is not installed, to the obvious plan: what the constraint asks for, row-wise by default)."""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field, fields
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from .._torchrec import (HAVE_TORCHREC, EmbeddingComputeKernel, EmbeddingModuleShardingPlan, EnumerableShardingSpec,
                         ParameterConstraints, ParameterSharding, ShardingPlan, ShardingType, ShardMetadata, Topology,
                         get_local_size, placement)
from ..dynamicemb_config import (DEFAULT_INDEX_TYPE, DynamicEmbKernel, DynamicEmbTableOptions, _sharded_table_bucket_layout,
                                 align_to_table_size, complete_initializer_args, data_type_to_dtype)

HBM_CAP: int = 288 * 1024 ** 3     # an MI355X
DDR_CAP: int = 512 * 1024 ** 3


@dataclass
class DynamicEmbParameterConstraints(ParameterConstraints):
    """ParameterConstraints + `use_dynamicemb` (store this table in a dynamic embedding table) and its options"""
    use_dynamicemb: Optional[bool] = False
    dynamicemb_options: Optional[DynamicEmbTableOptions] = field(default_factory=DynamicEmbTableOptions)


@dataclass
class DynamicEmbParameterSharding(ParameterSharding):
    """plan entry of a dynamic table; the extra fields travel to the compute kernel inside `fused_params`"""
    compute_kernel: str = EmbeddingComputeKernel.CUSTOMIZED_KERNEL.value
    customized_compute_kernel: Optional[str] = DynamicEmbKernel
    dist_type: str = "roundrobin"
    dynamicemb_options: Optional[DynamicEmbTableOptions] = field(default_factory=DynamicEmbTableOptions)

    _EXTRA = ("customized_compute_kernel", "dist_type", "dynamicemb_options")

    def get_additional_fused_params(self) -> Dict[str, Any]:
        base = {f.name for f in fields(ParameterSharding)}
        return {f.name: getattr(self, f.name) for f in fields(DynamicEmbParameterSharding) if f.name not in base}

    @staticmethod
    def pop_additional_fused_params(fused_params: Dict[str, Any]) -> None:
        """the planner-only keys are not keyword arguments of BatchedDynamicEmbeddingTablesV2"""
        for k in DynamicEmbParameterSharding._EXTRA:
            fused_params.pop(k, None)


def _prepare_dynemb_table_options(constraints: Dict[str, DynamicEmbParameterConstraints], eb_configs: List[Any],
                                  world_size: Optional[int] = None) -> None:
    """Checks that constraints and table configs name the same tables, then turns the GLOBAL options of every dynamic
    table into this rank's: initializer bounds, bucket layout and per-rank capacity, HBM budget / W, init capacity aligned
    and clamped, dim / index type / embedding dtype from the table config."""
    if constraints is None or eb_configs is None:
        raise ValueError("Constraints and eb_configs must not be None")
    if world_size is None:
        world_size = dist.get_world_size()
    names = [c.name for c in eb_configs]
    for n in names:
        if n not in constraints:
            raise ValueError(f"Config name '{n}' does not match any key in constraints")
    if len(set(names)) != len(names):
        raise ValueError("Config names must be unique")
    if set(names) != set(constraints.keys()):
        raise ValueError("Not all constraint keys have matching BaseEmbeddingConfig names")
    for cfg in eb_configs:
        c = constraints[cfg.name]
        if not c.use_dynamicemb:
            continue
        o = c.dynamicemb_options
        o.initializer_args = complete_initializer_args(o.initializer_args, embedding_config=cfg)
        nb, width = _sharded_table_bucket_layout(cfg, world_size, o.bucket_capacity)
        o.bucket_capacity = width
        o.max_capacity = nb * width
        o.local_hbm_for_values = math.ceil(o.global_hbm_for_values / world_size)
        if o.init_capacity is not None:
            aligned = align_to_table_size(o.init_capacity, width)
            if aligned != o.init_capacity:
                warnings.warn(f"init_capacity is aligned to {aligned} from {o.init_capacity} (bucket_capacity={width})", UserWarning)
            if aligned > o.max_capacity:
                warnings.warn(f"init_capacity {aligned} exceeds max_capacity {o.max_capacity}; clamping init_capacity to "
                              "max_capacity", UserWarning)
                aligned = o.max_capacity
            o.init_capacity = aligned
        else:
            o.init_capacity = o.max_capacity
        o.dim = cfg.embedding_dim
        if o.index_type is None:
            o.index_type = DEFAULT_INDEX_TYPE
        if o.embedding_dtype is None:
            o.embedding_dtype = data_type_to_dtype(cfg.data_type)


def _row_wise_spec(rows_per_rank: int, dim: int, world_size: int, compute_device: str, local_size: int):
    return EnumerableShardingSpec([ShardMetadata(shard_sizes=[rows_per_rank, dim], shard_offsets=[rows_per_rank * r, 0],
                                                 placement=placement(compute_device, r, local_size)) for r in range(world_size)])


class DynamicEmbeddingShardingPlanner:
    """Same constructor as the reference's wrapper of TorchRec's EmbeddingShardingPlanner plus `eb_configs`, the table
    configs of the model (they carry num_embeddings / embedding_dim, which a constraint does not)."""

    def __init__(self, eb_configs: List[Any], topology: Optional[Topology] = None, batch_size: Optional[int] = None,
                 enumerator: Any = None, storage_reservation: Any = None, proposer: Any = None, partitioner: Any = None,
                 performance_model: Any = None, stats: Any = None,
                 constraints: Optional[Dict[str, DynamicEmbParameterConstraints]] = None, debug: bool = True) -> None:
        world_size = dist.get_world_size() if dist.is_initialized() else (topology.world_size if topology else 1)
        _prepare_dynemb_table_options(constraints, eb_configs, world_size)
        self._constraints = constraints
        self._configs = {c.name: c for c in eb_configs}
        if topology is None:
            warnings.warn("No topology provided: the memory model of the planner falls back to one MI355X per rank "
                          "(288 GB HBM, 512 GB DDR).", RuntimeWarning)
            topology = Topology(local_world_size=get_local_size(world_size), world_size=world_size,
                                compute_device="cuda" if torch.cuda.is_available() else "cpu", hbm_cap=HBM_CAP, ddr_cap=DDR_CAP)
        self._topology = topology
        static = {k: c for k, c in constraints.items() if not c.use_dynamicemb}
        self._torchrec_planner = None
        if HAVE_TORCHREC:  # pragma: no cover - only where torchrec is installed
            from torchrec.distributed.planner import EmbeddingShardingPlanner

            self._torchrec_planner = EmbeddingShardingPlanner(
                topology=topology, batch_size=batch_size, enumerator=enumerator, storage_reservation=storage_reservation,
                proposer=proposer, partitioner=partitioner, performance_model=performance_model, stats=stats,
                constraints=static, debug=debug)
        local = topology.local_world_size or world_size
        self._dyn_emb_plan: Dict[str, DynamicEmbParameterSharding] = {}
        for name, c in constraints.items():
            if not c.use_dynamicemb:
                continue
            o = c.dynamicemb_options
            self._dyn_emb_plan[name] = DynamicEmbParameterSharding(
                sharding_type=ShardingType.ROW_WISE.value, compute_kernel=EmbeddingComputeKernel.CUSTOMIZED_KERNEL.value,
                ranks=list(range(world_size)),
                sharding_spec=_row_wise_spec(o.max_capacity, self._configs[name].embedding_dim, world_size,
                                             topology.compute_device, local),
                customized_compute_kernel=DynamicEmbKernel, dist_type=o.dist_type, dynamicemb_options=o)
        self._world_size = world_size

    # -- the plan of everything that is not a dynamic table, without TorchRec: what the constraint asks for ----------------
    def _static_entry(self, cfg, c: Optional[ParameterConstraints]) -> ParameterSharding:
        st = (c.sharding_types[0] if c is not None and c.sharding_types else ShardingType.ROW_WISE.value)
        ck = (c.compute_kernels[0] if c is not None and c.compute_kernels else
              (EmbeddingComputeKernel.DENSE.value if st == ShardingType.DATA_PARALLEL.value else EmbeddingComputeKernel.FUSED.value))
        W = self._world_size
        spec = None
        if st == ShardingType.ROW_WISE.value:
            spec = _row_wise_spec(-(-cfg.num_embeddings // W), cfg.embedding_dim, W, self._topology.compute_device,
                                  self._topology.local_world_size or W)
        return ParameterSharding(sharding_type=st, compute_kernel=ck, ranks=list(range(W)), sharding_spec=spec)

    def _module_configs(self, m: nn.Module):
        get = getattr(m, "embedding_bag_configs", None) or getattr(m, "embedding_configs", None)
        return list(get()) if get is not None else []

    def collective_plan(self, module: nn.Module, sharders: List[Any], pg: Optional[dist.ProcessGroup] = None) -> ShardingPlan:
        """the plan of every shardable module of `module`: TorchRec's for the static tables, the fixed row-wise entry
        for the dynamic ones.  (Every rank computes the same plan from the same inputs; no broadcast is needed for the
        dynamic entries.)"""
        if self._torchrec_planner is not None:  # pragma: no cover
            plan = self._torchrec_planner.collective_plan(module, sharders, pg if pg is not None else dist.GroupMember.WORLD)
            for mplan in plan.plan.values():
                for table in list(mplan.keys()):
                    if table in self._dyn_emb_plan:
                        mplan[table] = self._dyn_emb_plan[table]
            return plan
        types = {s.module_type for s in sharders}
        plan: Dict[str, EmbeddingModuleShardingPlan] = {}
        for path, m in module.named_modules():
            if type(m) not in types:
                continue
            mp = EmbeddingModuleShardingPlan()
            for cfg in self._module_configs(m):
                mp[cfg.name] = self._dyn_emb_plan.get(cfg.name) or self._static_entry(cfg, self._constraints.get(cfg.name))
            plan[path] = mp
        return ShardingPlan(plan)

    def plan(self, module: nn.Module, sharders: List[Any]) -> ShardingPlan:
        return self.collective_plan(module, sharders, None)
