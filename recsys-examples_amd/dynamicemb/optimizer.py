"""Optimizer names of the plugin surface (reference dynamicemb/optimizer.py:36-118): the optimizer maths itself lives in
the fused backward kernel (csrc/backward.hip), these are the argument carriers the TorchRec side passes around."""
from dataclasses import dataclass
from typing import Any, Dict

from .dynamicemb_config import EmbOptimType, get_optimizer_state_dim  # noqa: F401


def get_optimizer_ckpt_state_dim(optimizer_type: EmbOptimType, dim: int) -> int:
    """optimizer-state elements per row in a checkpoint file (the row-wise state is stored unpadded: one element)"""
    return {"SGD": 0, "EXACT_SGD": 0, "ADAM": 2 * dim, "EXACT_ADAGRAD": dim, "EXACT_ROWWISE_ADAGRAD": 1}[optimizer_type.name]


@dataclass
class OptimizerArgs:
    """the FBGEMM-TBE optimizer keyword set (only the first block is used by the supported optimizers)"""
    stochastic_rounding: bool = True
    gradient_clipping: bool = False
    max_gradient: float = 1.0
    max_norm: float = 0.0
    learning_rate: float = 0.01
    eps: float = 1.0e-8
    initial_accumulator_value: float = 0.0
    beta1: float = 0.9
    beta2: float = 0.999
    weight_decay: float = 0.0
    weight_decay_mode: int = 0
    eta: float = 0.001
    momentum: float = 0.9
    counter_halflife: int = -1
    adjustment_iter: int = -1
    adjustment_ub: float = 1.0
    learning_rate_mode: int = -1
    grad_sum_decay: int = -1
    tail_id_threshold: float = 0
    is_tail_id_thresh_ratio: int = 0
    total_hash_size: int = 0
    weight_norm_coefficient: float = 0
    lower_bound: float = 0
    regularization_mode: int = 0


def string_to_opt_type(optimizer_str: str) -> EmbOptimType:
    try:
        return EmbOptimType(optimizer_str)
    except ValueError:
        raise ValueError(f"'{optimizer_str}' is not a valid EmbOptimType.")


def get_required_arg(args: Dict[str, Any], key: str) -> Any:
    if key not in args:
        raise ValueError(f"Input args does not contain required optimizer argument: {key}")
    return args[key]


class OptimizerView:
    """What an external `Storage` is told about the optimizer of its tables (the reference constructs it with the
    BaseDynamicEmbeddingOptimizer object, optimizer.py:118-190; a store only needs the row layout and the checkpoint
    metadata of it): state columns per embedding dim, their initial value, the hyper-parameters."""

    def __init__(self, module):
        self._m = module

    def get_state_dim(self, emb_dim: int) -> int:
        return get_optimizer_state_dim(self._m.optimizer_type, emb_dim, self._m.embedding_dtype)

    def get_ckpt_state_dim(self, emb_dim: int) -> int:
        return get_optimizer_ckpt_state_dim(self._m.optimizer_type, emb_dim)

    def get_initial_optim_states(self) -> float:
        return float(self._m.initial_accumulator_value)

    def get_opt_args(self) -> Dict[str, Any]:
        return self._m._opt_args()

    def set_learning_rate(self, new_lr: float) -> None:
        self._m.learning_rate = new_lr
