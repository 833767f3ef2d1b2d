"""DynamicEmb for MI355X: the package surface of the reference's `dynamicemb` (corelib/dynamicemb/dynamicemb/__init__.py:16-75)
-- table options and enums, admission, optimizer argument types, dump / load -- over the gfx950 kernels of librecsys_amd.so.
The TorchRec plugin pieces live in the submodules the reference uses: `dynamicemb.shard` (sharders),
`dynamicemb.planner` (planner, constraints, enumerator), `dynamicemb.get_planner`, `dynamicemb.utils` (TORCHREC_TYPES),
`dynamicemb.batched_dynamicemb_compute_kernel`."""
from .dump_load import DynamicEmbDump, DynamicEmbLoad
from .dynamicemb_config import (BATCH_SIZE_PER_DUMP, BUCKET_ALIGNMENT, DEMB_TABLE_ALIGN_SIZE, MAX_BUCKET_CAPACITY,
                                DynamicEmbCheckMode, DynamicEmbEvictStrategy, DynamicEmbInitializerArgs,
                                DynamicEmbInitializerMode, DynamicEmbPoolingMode, DynamicEmbScoreStrategy,
                                DynamicEmbTableOptions, EmbOptimType, ScoreStrategy, align_to_table_size, data_type_to_dtype,
                                data_type_to_dyn_emb, dyn_emb_to_torch, get_sharded_table_capacity, get_table_value_bytes,
                                string_to_evict_strategy)
from .embedding_admission import AdmissionStrategy, Counter, FrequencyAdmissionStrategy, KVCounter
from .optimizer import OptimizerArgs
from .utils import torch_to_dyn_emb

__all__ = [
    "AdmissionStrategy", "BUCKET_ALIGNMENT", "DEMB_TABLE_ALIGN_SIZE", "MAX_BUCKET_CAPACITY", "align_to_table_size",
    "get_table_value_bytes", "get_sharded_table_capacity", "FrequencyAdmissionStrategy", "Counter", "KVCounter",
    "DynamicEmbCheckMode", "DynamicEmbInitializerArgs", "DynamicEmbInitializerMode", "DynamicEmbTableOptions",
    "DynamicEmbPoolingMode", "DynamicEmbEvictStrategy", "DynamicEmbScoreStrategy", "ScoreStrategy", "BATCH_SIZE_PER_DUMP",
    "data_type_to_dyn_emb", "data_type_to_dtype", "dyn_emb_to_torch", "torch_to_dyn_emb", "string_to_evict_strategy",
    "DynamicEmbDump", "DynamicEmbLoad", "EmbOptimType", "OptimizerArgs",
]
