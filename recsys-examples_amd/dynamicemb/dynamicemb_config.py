"""Configuration surface of the DynamicEmb plugin, mirroring the reference's names
(corelib/dynamicemb/dynamicemb/dynamicemb_config.py:58-519, types.py:33-116) so that
`examples/commons/distributed/sharding.py` style call sites keep working."""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Any, Optional, Tuple, Union

import torch

from dynamicemb_extensions import DynamicEmbDataType, EvictStrategy  # noqa: F401

try:  # the real enum when fbgemm_gpu is installed
    from fbgemm_gpu.split_embedding_configs import EmbOptimType  # type: ignore
except Exception:  # pragma: no cover - fbgemm_gpu is absent in this image

    class EmbOptimType(enum.Enum):  # names as fbgemm_gpu.split_embedding_configs.EmbOptimType
        SGD = "sgd"
        EXACT_SGD = "exact_sgd"
        LAMB = "lamb"
        ADAM = "adam"
        EXACT_ADAGRAD = "exact_adagrad"
        EXACT_ROWWISE_ADAGRAD = "exact_row_wise_adagrad"
        LARS_SGD = "lars_sgd"
        PARTIAL_ROWWISE_ADAM = "partial_row_wise_adam"
        PARTIAL_ROWWISE_LAMB = "partial_row_wise_lamb"
        NONE = "none"


DEFAULT_INDEX_TYPE = torch.int64
SUPPORTED_DIST_TYPES = ("continuous", "roundrobin", "hash_roundrobin")
DEBUG_EMB_INITIALIZER_MOD = 100_000
DEFAULT_BUCKET_CAPACITY = 128
DynamicEmbKernel = "DynamicEmb"
KEY_TYPE = torch.int64
EMBEDDING_TYPE = torch.float32
SCORE_TYPE = torch.int64


@enum.unique
class DynamicEmbCheckMode(enum.IntEnum):
    ERROR = 0
    WARNING = 1
    IGNORE = 2


class DynamicEmbPoolingMode(enum.IntEnum):
    SUM = 0
    MEAN = 1
    NONE = 2


@enum.unique
class DynamicEmbEvictStrategy(enum.Enum):
    LRU = EvictStrategy.KLru
    LFU = EvictStrategy.KLfu
    EPOCH_LRU = EvictStrategy.KEpochLru
    EPOCH_LFU = EvictStrategy.KEpochLfu
    CUSTOMIZED = EvictStrategy.KCustomized


class DynamicEmbScoreStrategy(enum.IntEnum):
    TIMESTAMP = 0
    STEP = 1
    CUSTOMIZED = 2
    LFU = 3
    NO_EVICTION = 4


ScoreStrategy = Union[DynamicEmbScoreStrategy, Tuple[DynamicEmbScoreStrategy, ...]]


class DynamicEmbInitializerMode(enum.Enum):
    NORMAL = "normal"
    TRUNCATED_NORMAL = "truncated_normal"
    UNIFORM = "uniform"
    CONSTANT = "constant"
    DEBUG = "debug"


@dataclass
class DynamicEmbInitializerArgs:
    mode: DynamicEmbInitializerMode = DynamicEmbInitializerMode.UNIFORM
    mean: float = 0.0
    std_dev: float = 1.0
    lower: float = None
    upper: float = None
    value: float = 0.0


def normalize_score_strategy(s):
    if s is None:
        return None
    if isinstance(s, tuple):
        if len(s) == 1:
            return s[0]
        if frozenset(s) == frozenset({DynamicEmbScoreStrategy.TIMESTAMP, DynamicEmbScoreStrategy.LFU}) and len(s) == 2:
            return tuple(s)
        raise NotImplementedError(f"Unsupported compound score_strategy {s}.")
    if not isinstance(s, DynamicEmbScoreStrategy):
        raise TypeError("score_strategy must be a DynamicEmbScoreStrategy or a tuple of them")
    return s


@dataclass
class DynamicEmbTableOptions:
    """Same fields and defaults as the reference dataclass (dynamicemb_config.py:308-519)."""

    embedding_dtype: Optional[torch.dtype] = None
    dim: Optional[int] = None
    max_capacity: Optional[int] = None
    evict_strategy: DynamicEmbEvictStrategy = DynamicEmbEvictStrategy.LRU
    local_hbm_for_values: int = 0
    device_id: Optional[int] = None
    training: bool = True
    initializer_args: DynamicEmbInitializerArgs = field(default_factory=DynamicEmbInitializerArgs)
    eval_initializer_args: DynamicEmbInitializerArgs = field(
        default_factory=lambda: DynamicEmbInitializerArgs(mode=DynamicEmbInitializerMode.CONSTANT, value=0.0))
    caching: bool = False
    init_capacity: Optional[int] = None
    max_load_factor: float = 0.5
    score_strategy: Optional[ScoreStrategy] = DynamicEmbScoreStrategy.TIMESTAMP
    bucket_capacity: int = DEFAULT_BUCKET_CAPACITY
    safe_check_mode: DynamicEmbCheckMode = DynamicEmbCheckMode.IGNORE
    global_hbm_for_values: int = 0
    external_storage: Any = None
    index_type: Optional[torch.dtype] = None
    dist_type: str = "roundrobin"
    admit_strategy: Any = None
    admission_counter: Any = None

    def __post_init__(self):
        assert self.eval_initializer_args.mode == DynamicEmbInitializerMode.CONSTANT, \
            "eval_initializer_args must be constant initialization"
        if self.dist_type not in SUPPORTED_DIST_TYPES:
            raise ValueError(f"Unsupported dist_type {self.dist_type!r}. Supported values: {SUPPORTED_DIST_TYPES}.")
        self.score_strategy = normalize_score_strategy(self.score_strategy)

    def get_grouped_key(self):
        return {"training": self.training, "caching": self.caching, "external_storage": self.external_storage,
                "index_type": self.index_type, "dist_type": self.dist_type, "score_strategy": self.score_strategy,
                "admit_strategy": self.admit_strategy}

    def __eq__(self, other):
        if not isinstance(other, DynamicEmbTableOptions):
            return NotImplemented
        return self.get_grouped_key() == other.get_grouped_key()

    def __ne__(self, other):
        return not (self == other)

    def __hash__(self):
        return hash(tuple(self.get_grouped_key().items()))


def dtype_to_bytes(dtype: torch.dtype) -> int:
    return torch.empty(0, dtype=dtype).element_size()


def get_optimizer_state_dim(optimizer: "EmbOptimType", dim: int, dtype: torch.dtype) -> int:
    """Elements of optimizer state appended to every row (optimizer.py:36-58)."""
    name = optimizer.name
    if name in ("SGD", "EXACT_SGD"):
        return 0
    if name == "ADAM":
        return 2 * dim
    if name == "EXACT_ADAGRAD":
        return dim
    if name == "EXACT_ROWWISE_ADAGRAD":
        return 16 // dtype_to_bytes(dtype)
    raise ValueError(f"Not supported optimizer type: {optimizer}")


# ---- helpers of the plugin surface (dynamicemb_config.py:522-800 of the reference) -------------------------------------------
BATCH_SIZE_PER_DUMP = 65536
DEMB_TABLE_ALIGN_SIZE = 16                 # types.py:113
BUCKET_ALIGNMENT: int = DEMB_TABLE_ALIGN_SIZE
MAX_BUCKET_CAPACITY: int = 2 ** 63 - 1     # "one bucket spans the whole per-rank shard"

_DTYPE_BY_NAME = {"FP32": torch.float32, "FP16": torch.float16, "BF16": torch.bfloat16}
_DYN_BY_NAME = {"FP32": DynamicEmbDataType.Float32, "FP16": DynamicEmbDataType.Float16, "BF16": DynamicEmbDataType.BFloat16}


def data_type_to_dtype(data_type) -> torch.dtype:
    """TorchRec DataType -> torch dtype (FP32 / FP16 / BF16: the types a dynamic table can hold)"""
    try:
        return _DTYPE_BY_NAME[data_type.name]
    except (KeyError, AttributeError):
        raise ValueError(f"DataType {data_type} cannot be converted to a torch dtype of a dynamic embedding table")


def data_type_to_dyn_emb(data_type) -> DynamicEmbDataType:
    try:
        return _DYN_BY_NAME[data_type.name]
    except (KeyError, AttributeError):
        raise ValueError(f"DataType {data_type} cannot be converted to DynamicEmbDataType")


def dyn_emb_to_torch(data_type: DynamicEmbDataType) -> torch.dtype:
    table = {DynamicEmbDataType.Float32: torch.float32, DynamicEmbDataType.BFloat16: torch.bfloat16,
             DynamicEmbDataType.Float16: torch.float16, DynamicEmbDataType.Int64: torch.int64,
             DynamicEmbDataType.UInt64: torch.uint64, DynamicEmbDataType.Int32: torch.int32,
             DynamicEmbDataType.UInt32: torch.uint32, DynamicEmbDataType.Size_t: torch.int64}
    if data_type not in table:
        raise ValueError(f"Unsupported DynamicEmbDataType: {data_type}")
    return table[data_type]


def string_to_evict_strategy(strategy_str: str) -> EvictStrategy:
    try:
        return {e.name: e for e in EvictStrategy}[strategy_str]
    except KeyError:
        raise ValueError(f"Invalid EvictStrategy string: {strategy_str}")


def get_eviction_score_strategy(score_strategy: ScoreStrategy) -> DynamicEmbScoreStrategy:
    """the strategy whose score column drives eviction: the LFU half of (TIMESTAMP, LFU), else the strategy itself"""
    s = normalize_score_strategy(score_strategy)
    if isinstance(s, tuple):
        return DynamicEmbScoreStrategy.LFU
    return s


def complete_initializer_args(initializer_args: DynamicEmbInitializerArgs, *, embedding_config=None) -> DynamicEmbInitializerArgs:
    """missing UNIFORM bounds default to +-sqrt(1 / num_embeddings) of the table (0 / 1 without a config); returns a new
    object when something was filled in, the argument itself otherwise"""
    from dataclasses import replace

    if initializer_args.mode != DynamicEmbInitializerMode.UNIFORM:
        return initializer_args
    if initializer_args.lower is not None and initializer_args.upper is not None:
        return initializer_args
    if embedding_config is not None:
        s = (1.0 / float(embedding_config.num_embeddings)) ** 0.5
        lo, hi = -s, s
    else:
        lo, hi = 0.0, 1.0
    return replace(initializer_args, lower=lo if initializer_args.lower is None else initializer_args.lower,
                   upper=hi if initializer_args.upper is None else initializer_args.upper)


def align_to_table_size(n: int, alignment: int = DEMB_TABLE_ALIGN_SIZE) -> int:
    """n rounded up to a multiple of `alignment`; anything <= 0 becomes one alignment unit (no zero-capacity tables)"""
    n = int(n)
    return alignment if n <= 0 else -(-n // alignment) * alignment


def _sharded_table_bucket_layout(embedding_config, world_size: int, bucket_capacity: int) -> Tuple[int, int]:
    """(number of buckets, bucket width in rows) of ONE rank's shard of a row-wise sharded table"""
    if world_size <= 0:
        raise ValueError(f"world_size must be positive, got {world_size}")
    shard_rows = -(-int(embedding_config.num_embeddings) // world_size)
    if bucket_capacity == MAX_BUCKET_CAPACITY:
        return 1, align_to_table_size(shard_rows, BUCKET_ALIGNMENT)
    if bucket_capacity <= 0:
        raise ValueError(f"bucket_capacity must be positive when not MAX_BUCKET_CAPACITY, got {bucket_capacity}")
    if bucket_capacity % BUCKET_ALIGNMENT != 0:
        raise ValueError(f"bucket_capacity ({bucket_capacity}) must be a multiple of BUCKET_ALIGNMENT ({BUCKET_ALIGNMENT}) "
                         "when not using MAX_BUCKET_CAPACITY.")
    return align_to_table_size(shard_rows, bucket_capacity) // bucket_capacity, bucket_capacity


def get_sharded_table_capacity(embedding_config, world_size: int, bucket_capacity: int) -> int:
    """rows one rank allocates for its shard: ceil(N / W) rounded up to whole buckets -- what the planner writes into
    DynamicEmbTableOptions.max_capacity"""
    nb, width = _sharded_table_bucket_layout(embedding_config, world_size, bucket_capacity)
    return int(nb * width)


def get_table_value_bytes(embedding_config, optimizer_type: "EmbOptimType", world_size: int,
                          bucket_capacity: int = DEFAULT_BUCKET_CAPACITY) -> int:
    """bytes of embedding + optimizer-state storage of one table over all ranks"""
    rows = get_sharded_table_capacity(embedding_config, world_size, bucket_capacity) * world_size
    dtype = data_type_to_dtype(embedding_config.data_type)
    dim = embedding_config.embedding_dim
    return rows * (dim + get_optimizer_state_dim(optimizer_type, dim, dtype)) * dtype_to_bytes(dtype)


def get_constraint_capacity(memory_bytes, dtype, dim, optimizer_type: "EmbOptimType", bucket_capacity) -> int:
    """rows (whole buckets, at least one) that fit into `memory_bytes` of value storage"""
    import warnings

    row = (dim + get_optimizer_state_dim(optimizer_type, dim, dtype)) * dtype_to_bytes(dtype)
    if memory_bytes < bucket_capacity * row:
        warnings.warn(f"Reserved HBM ({memory_bytes} bytes) is less than one bucket ({bucket_capacity * row} bytes). "
                      "Rounding up to one bucket.", UserWarning)
        memory_bytes = bucket_capacity * row
    return memory_bytes // row // bucket_capacity * bucket_capacity
