"""`dynamicemb.get_planner.get_planner` (reference get_planner.py:59-131): constraints for the three kinds of table a
model holds -- data-parallel (small, replicated), dynamic (row-wise, DynamicEmb kernel) and static model-parallel
(row-wise, TorchRec kernel) -- and the planner over them.  Bandwidth defaults are MI355X's: one xGMI link per GPU pair
(~153 GB/s per direction, all-to-all uses the 7 links of a GPU in parallel) instead of the reference's NVLink figure."""
from typing import Dict, List, Set

import torch
import torch.distributed as dist

from ._torchrec import BoundsCheckMode, HeuristicalStorageReservation, ShardingType, Topology, get_local_size
from .dynamicemb_config import DynamicEmbTableOptions
from .planner import DynamicEmbeddingEnumerator, DynamicEmbeddingShardingPlanner, DynamicEmbParameterConstraints

# which TorchRec compute kernels a train-pipeline flavour allows for the static tables
_MODEL_PARALLEL_KERNELS = {"prefetch": ["fused_uvm_caching"], "native": ["fused", "fused_uvm"], "none": []}
_DATA_PARALLEL_KERNELS = {"prefetch": ["dense"], "native": ["dense"], "none": []}


def get_planner(eb_configs: List, data_parallel_embedding_table_names: Set[str],
                dynamicemb_options_dict: Dict[str, DynamicEmbTableOptions], device: torch.device,
                pipeline_type: str = "none", ddr_cap: int = 512 * 1024 ** 3, intra_host_bw: int = 7 * 153e9,
                inter_host_bw: int = 50e9):
    if pipeline_type not in _MODEL_PARALLEL_KERNELS:
        raise ValueError(f"unknown pipeline_type {pipeline_type!r}")
    constraints = {}
    for cfg in eb_configs:
        if cfg.name in data_parallel_embedding_table_names:
            c = DynamicEmbParameterConstraints(sharding_types=[ShardingType.DATA_PARALLEL.value],
                                               bounds_check_mode=BoundsCheckMode.NONE, use_dynamicemb=False,
                                               compute_kernels=_DATA_PARALLEL_KERNELS[pipeline_type])
        elif cfg.name in dynamicemb_options_dict:
            c = DynamicEmbParameterConstraints(sharding_types=[ShardingType.ROW_WISE.value],
                                               bounds_check_mode=BoundsCheckMode.NONE,   # a dynamic table has no bounds
                                               enforce_hbm=True, use_dynamicemb=True,
                                               dynamicemb_options=dynamicemb_options_dict[cfg.name])
        else:
            c = DynamicEmbParameterConstraints(sharding_types=[ShardingType.ROW_WISE.value],
                                               bounds_check_mode=BoundsCheckMode.NONE, use_dynamicemb=False,
                                               compute_kernels=_MODEL_PARALLEL_KERNELS[pipeline_type])
        constraints[cfg.name] = c
    hbm_cap = torch.cuda.get_device_properties(0).total_memory if torch.cuda.is_available() else 288 * 1024 ** 3
    world = dist.get_world_size() if dist.is_initialized() else 1
    topology = Topology(local_world_size=get_local_size(world), world_size=world, compute_device=device.type, hbm_cap=hbm_cap,
                        ddr_cap=ddr_cap, intra_host_bw=intra_host_bw, inter_host_bw=inter_host_bw)
    enumerator = DynamicEmbeddingEnumerator(topology=topology, constraints=constraints)
    return DynamicEmbeddingShardingPlanner(eb_configs=eb_configs, topology=topology, constraints=constraints,
                                           enumerator=enumerator,
                                           storage_reservation=HeuristicalStorageReservation(percentage=0.05))
