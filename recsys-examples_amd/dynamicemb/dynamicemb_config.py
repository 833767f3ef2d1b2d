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
