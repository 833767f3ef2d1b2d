"""Admission control: which unseen keys may enter the embedding table.

Mirror of the reference's `dynamicemb/embedding_admission.py` (KVCounter :35-50, MultiTableKVCounter :53-104,
FrequencyAdmissionStrategy :107-180) and of the `Counter` / `AdmissionStrategy` interfaces (`types.py:333-420`).
The counter is one fused scored hash table (ACCUMULATE policy) over all logical tables, i.e. the same HIP kernels as
the embedding table's key index; `add()` is its insert with `score_out`.
"""
from __future__ import annotations

import abc
from typing import List, Optional

import torch

from .dynamicemb_config import DynamicEmbInitializerArgs
from .scored_hashtable import LinearBucketTable, ScoreArg, ScorePolicy, ScoreSpec


class Counter(abc.ABC):
    """key -> counter map, multi-table through `table_ids` (types.py:333-399)."""

    @abc.abstractmethod
    def add(self, keys: torch.Tensor, table_ids: torch.Tensor, frequencies: torch.Tensor) -> torch.Tensor:
        """add `frequencies` to the (unique) keys' counters and return the accumulated values"""

    @abc.abstractmethod
    def erase(self, keys: torch.Tensor, table_ids: torch.Tensor) -> None:
        ...

    @abc.abstractmethod
    def memory_usage(self, mem_type=None) -> int:
        ...


class AdmissionStrategy(abc.ABC):
    """types.py:401-420"""

    @abc.abstractmethod
    def admit(self, keys: torch.Tensor, frequencies: torch.Tensor) -> torch.Tensor:
        """boolean mask of the keys that may enter the table"""

    @abc.abstractmethod
    def initialize_non_admitted_embeddings(self, buffer: torch.Tensor, indices: torch.Tensor) -> bool:
        ...


class KVCounter:
    """Per-table counter configuration (embedding_admission.py:35-50)."""

    def __init__(self, capacity: int, bucket_capacity: int = 1024, key_type: torch.dtype = torch.int64):
        self.capacity = capacity
        self.bucket_capacity = bucket_capacity
        self.key_type = key_type


class MultiTableKVCounter(Counter):
    """One fused counter table for a list of per-table configs (embedding_admission.py:53-104)."""

    def __init__(self, kv_counters: List[KVCounter], device: torch.device):
        if not kv_counters:
            raise ValueError("kv_counters must be non-empty")
        self.score_name_ = "counter"
        self.score_specs_ = [ScoreSpec(self.score_name_, ScorePolicy.ACCUMULATE)]
        self.table_ = LinearBucketTable([kv.capacity for kv in kv_counters], self.score_specs_,
                                        key_type=kv_counters[0].key_type, bucket_capacity=kv_counters[0].bucket_capacity,
                                        device=device)

    def add(self, keys: torch.Tensor, table_ids: torch.Tensor, frequencies: torch.Tensor) -> torch.Tensor:
        scores_out = torch.empty(keys.numel(), dtype=torch.int64, device=keys.device)
        if keys.numel():
            self.table_.insert(keys, table_ids, ScoreArg(self.score_name_, frequencies.to(torch.int64).contiguous()),
                               score_out=scores_out)
        return scores_out

    def erase(self, keys: torch.Tensor, table_ids: torch.Tensor) -> None:
        if keys.numel():
            self.table_.erase(keys, table_ids)

    def memory_usage(self, mem_type=None) -> int:
        return self.table_.memory_usage(mem_type)

    def size(self):
        return self.table_.size()


class FrequencyAdmissionStrategy(AdmissionStrategy):
    """Admit a key once its accumulated frequency reaches `threshold` (embedding_admission.py:107-180).
    `initializer_args`: how the embedding of a key that is NOT admitted is produced (None: the table's initializer)."""

    def __init__(self, threshold: int, initializer_args: Optional[DynamicEmbInitializerArgs] = None):
        if threshold < 0:
            raise ValueError(f"Threshold must be non-negative, got {threshold}")
        self.threshold = threshold
        self.initializer_args = initializer_args

    def admit(self, keys: torch.Tensor, frequencies: torch.Tensor) -> torch.Tensor:
        if keys.shape[0] != frequencies.shape[0]:
            raise ValueError(f"Keys and frequencies must have same length, got {keys.shape[0]} and {frequencies.shape[0]}")
        return frequencies >= self.threshold

    def initialize_non_admitted_embeddings(self, buffer: torch.Tensor, indices: torch.Tensor) -> bool:
        """Dense-buffer form of the reference interface: rows `indices` of `buffer` get the strategy's initializer
        (False: no initializer configured -- the caller falls back to the table's).  The module itself initialises the
        scratch rows of non-admitted keys with init_rows (same counter-based generators, keyed by the key)."""
        if self.initializer_args is None:
            return False
        from .batched_dynamicemb_tables import init_dense_rows

        init_dense_rows(buffer, indices, self.initializer_args)
        return True
