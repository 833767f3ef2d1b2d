"""Host-side owner of the scored hash table: mirrors the interface of the reference's
``LinearBucketTable`` / ``ScoredHashTable`` (corelib/dynamicemb/dynamicemb/scored_hashtable.py:83-291,
294-1757) -- same constructor arguments, method names, return tuples and DEMB_DETERMINISM_MODE
behaviour -- on top of the gfx950 kernels (dynamicemb_extensions -> librecsys_amd.so).

State is plain tensors owned here (one uint8 arena: per bucket [keys u64 x C][digests u8 x C]
[scores u64 x C x num_scores], bucket_sizes i32, ref counter i32, bucket offsets i64), exactly the
layout the reference documents, so dumps of the arena are interchangeable.
Overflow buckets of cache tables (enable_overflow=True, scored_hashtable.py:426-474 of the reference) are built;
dump / load / incremental_dump live at the module level (batched_dynamicemb_tables.py).
"""
from __future__ import annotations

import abc
import enum
import os
import warnings
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

import dynamicemb_extensions as ext
from dynamicemb_extensions import ScorePolicy


@dataclass(frozen=True)
class ScoreSpec:  # scored_hashtable.py:46-53
    name: str
    policy: ScorePolicy
    dtype: torch.dtype = torch.uint64
    priority: int = 0
    is_reduction: bool = True


@dataclass
class ScoreArg:  # scored_hashtable.py:56-62
    name: str
    value: Optional[torch.Tensor] = None
    policy: Optional[ScorePolicy] = None


def score_policy_num_scores(policy) -> int:  # scored_hashtable.py:65-69
    return 2 if policy == ScorePolicy.LRU_LFU else 1


@enum.unique
class ProbingType(enum.Enum):
    LINEAR = "linear"
    CHAINED = "separate_chain"


@enum.unique
class ReductionType(enum.Enum):
    LINEAR = "linear"
    DOUBLY_LINKED = "doubly_linked"


class ScoredHashTable(abc.ABC):
    """Abstract interface (scored_hashtable.py:83-273)."""

    @property
    def index_type(self) -> torch.dtype:
        return torch.int64

    @property
    def result_type(self) -> torch.dtype:
        return torch.uint8


class LinearBucketTable(ScoredHashTable):
    def __init__(self, capacity: List[int], score_specs: List[ScoreSpec], key_type: torch.dtype = torch.int64,
                 bucket_capacity: Optional[int] = None, device: torch.device = None, enable_overflow: bool = False,
                 host: bool = False):
        """host=True keeps the table (keys / digests / scores, bucket sizes, pin counters) in pinned host memory, which
        the GPU kernels address directly (the host tier of the reference's HybridStorage / host-only storage,
        key_value_table.py:2107-2403; the reference stages through HostVMMTensor)."""
        assert not (enable_overflow and host), "the overflow region belongs to the HBM cache table"
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        assert key_type in (torch.int64, torch.uint64), "Only accept 64 bits integer as key's type."
        self.key_type_ = key_type
        assert len(score_specs) == 1, "Only a single ScoreSpec is supported."
        self.score_specs_ = list(score_specs)
        self.score_names_ = [s.name for s in self.score_specs_]
        self.num_scores_ = sum(score_policy_num_scores(s.policy) for s in self.score_specs_)
        if bucket_capacity is None:
            bucket_capacity = 128
        self.bucket_capacity_ = ((bucket_capacity + 15) // 16) * 16
        if self.bucket_capacity_ != bucket_capacity:
            warnings.warn(f"Bucket capacity is rounded from {bucket_capacity} to {self.bucket_capacity_}.", UserWarning)
        assert isinstance(capacity, list) and len(capacity) >= 1
        C = self.bucket_capacity_
        self.num_tables_ = len(capacity)
        self.per_table_num_buckets_ = [(c + C - 1) // C for c in capacity]
        self.per_table_capacity_ = [n * C for n in self.per_table_num_buckets_]
        offs = [0]
        for n in self.per_table_num_buckets_:
            offs.append(offs[-1] + n)
        self.num_buckets_ = offs[-1]
        self.capacity_ = self.num_buckets_ * C
        self.table_bucket_offsets_ = torch.tensor(offs, dtype=torch.int64, device=self.device)
        self.table_bucket_offsets_cpu_ = torch.tensor(offs, dtype=torch.int64)
        self.storage_bytes_ = (9 + 8 * self.num_scores_) * C * self.num_buckets_
        self.host_ = host
        if host:
            self.table_storage_ = torch.empty(self.storage_bytes_, dtype=torch.uint8, pin_memory=True)
            self.bucket_sizes = torch.zeros(self.num_buckets_, dtype=torch.int32).pin_memory()
            self._ref_counter = torch.zeros(self.capacity_, dtype=torch.int32).pin_memory()
        else:
            self.table_storage_ = torch.empty(self.storage_bytes_, dtype=torch.uint8, device=self.device)
            self.bucket_sizes = torch.zeros(self.num_buckets_, dtype=torch.int32, device=self.device)
            self._ref_counter = torch.zeros(self.capacity_, dtype=torch.int32, device=self.device)
        # overflow region (scored_hashtable.py:426-474): one bucket of 3*C slots per logical table in its own arena, its
        # ref-counters behind the main ones, indices of its entries offset by the table's main capacity
        self.enable_overflow_ = bool(enable_overflow)
        if self.enable_overflow_:
            self.overflow_bucket_capacity_ = 3 * C
            self.overflow_num_buckets_ = self.num_tables_
            self.overflow_table_storage_ = torch.empty(
                (9 + 8 * self.num_scores_) * self.overflow_bucket_capacity_ * self.overflow_num_buckets_, dtype=torch.uint8,
                device=self.device)
            self.overflow_bucket_sizes = torch.zeros(self.overflow_num_buckets_, dtype=torch.int32, device=self.device)
            self.overflow_output_offsets_ = torch.tensor(self.per_table_capacity_, dtype=torch.int64, device=self.device)
            self._ref_counter = torch.zeros(self.capacity_ + self.overflow_bucket_capacity_ * self.num_tables_,
                                            dtype=torch.int32, device=self.device)
            self._ovf_counter = self._ref_counter[self.capacity_:]
        else:
            self.overflow_bucket_capacity_ = 0
            self.overflow_num_buckets_ = 0
            self.overflow_table_storage_ = None
            self.overflow_bucket_sizes = None
            self.overflow_output_offsets_ = None
            self._ovf_counter = None
        self.reset()

    # -- views (table_partition) ---------------------------------------------------------
    def partition(self):
        dtypes = [self.key_type_, torch.uint8] + [torch.uint64] * self.num_scores_
        return ext.table_partition(self.table_storage_, [self.key_type_, torch.uint8, torch.uint64],
                                   self.bucket_capacity_, self.num_buckets_) if self.num_scores_ == 1 else None

    @property
    def key_type(self) -> torch.dtype:
        return self.key_type_

    @property
    def score_specs(self) -> List[ScoreSpec]:
        return self.score_specs_

    def _parse_score(self, score: ScoreArg):
        index = self.score_names_.index(score.name)
        policy = score.policy if score.policy is not None else self.score_specs_[index].policy
        return score.value, policy

    def reset(self) -> None:
        ext.table_init(self.table_storage_, self.bucket_capacity_, self.num_buckets_, self.num_scores_)
        self.bucket_sizes.zero_()
        self._ref_counter.zero_()
        if self.enable_overflow_:
            ext.table_init(self.overflow_table_storage_, self.overflow_bucket_capacity_, self.overflow_num_buckets_,
                           self.num_scores_)
            self.overflow_bucket_sizes.zero_()

    def _ovf(self):
        return dict(overflow_output_offsets=self.overflow_output_offsets_ if self.enable_overflow_ else None,
                    overflow_bucket_capacity=self.overflow_bucket_capacity_)

    # -- overflow (DynamicEmbCache, key_value_table.py:1522-1590) -----------------------------
    def lookup_with_overflow(self, keys, table_ids, score: ScoreArg):
        """scored_hashtable.py:736-763: lookup, then the table's overflow bucket for the keys the main table does not
        hold; overflow indices are offset by the per-table main capacity."""
        assert self.enable_overflow_, "lookup_with_overflow requires enable_overflow=True"
        value, policy = self._parse_score(score)
        return ext.table_lookup(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, keys, table_ids,
                                value, policy, ovf_storage=self.overflow_table_storage_,
                                ovf_bucket_capacity=self.overflow_bucket_capacity_,
                                ovf_output_offsets=self.overflow_output_offsets_, num_scores=self.num_scores_)

    def _insert_ovf(self, keys, table_ids, value, policy, insert_results=None, score_out=None):
        return ext.table_insert_and_evict(
            self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, self.bucket_sizes, keys, table_ids,
            value, policy, self._ref_counter, insert_results, score_out, ovf_storage=self.overflow_table_storage_,
            ovf_bucket_capacity=self.overflow_bucket_capacity_, ovf_bucket_sizes=self.overflow_bucket_sizes,
            ovf_counter=self._ovf_counter, ovf_output_offsets=self.overflow_output_offsets_, num_scores=self.num_scores_)

    def insert_and_evict_with_counter_and_overflow(self, keys, table_ids, score: ScoreArg, insert_results=None,
                                                   score_out=None):
        """scored_hashtable.py:765-824: counter-aware insert; keys whose bucket is entirely pinned go to the overflow
        bucket (victims there: ref-counter 0).  Same tuple as insert_and_evict."""
        assert self.enable_overflow_, "insert_and_evict_with_counter_and_overflow requires enable_overflow=True"
        value, policy = self._parse_score(score)
        if os.environ.get("DEMB_DETERMINISM_MODE") is not None:
            assert self.num_scores_ == 1, "DEMB_DETERMINISM_MODE does not support auxiliary score columns yet."
            return self._deterministic_insert_and_evict_with_overflow(keys, table_ids, value, policy)
        idx, nev, ek, ei, es, et = self._insert_ovf(keys, table_ids, value, policy, insert_results, score_out)
        h = int(nev.cpu().item())
        return idx, h, ek[:h], ei[:h], es[:h], et[:h]

    def _deterministic_insert_and_evict_with_overflow(self, keys, table_ids, score_value, policy):
        """scored_hashtable.py:1643-1735: bucket waves through the overflow insert, final indices by one CONST lookup"""
        n = keys.numel()
        dev = keys.device
        if n == 0:
            e = torch.empty(0, dtype=torch.int64, device=dev)
            return e, 0, torch.empty_like(keys[:0]), e.clone(), e.clone(), e.clone()
        acc = [[], [], [], []]
        for vk, vt, vs in self._waves(keys, table_ids, score_value):
            _, nev, ek, ei, es, et = self._insert_ovf(vk, vt, vs, policy)
            h = int(nev.cpu().item())
            if h:
                for lst, a in zip(acc, (ek, ei, es, et)):
                    lst.append(a[:h])
        _, _, indices = ext.table_lookup(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, keys,
                                         table_ids, None, ScorePolicy.CONST, ovf_storage=self.overflow_table_storage_,
                                         ovf_bucket_capacity=self.overflow_bucket_capacity_,
                                         ovf_output_offsets=self.overflow_output_offsets_)
        cat = [torch.cat(x) if x else torch.empty(0, dtype=(keys.dtype if i == 0 else torch.int64), device=dev)
               for i, x in enumerate(acc)]
        return indices, cat[0].numel(), cat[0], cat[1], cat[2], cat[3]

    # -- lookup / insert --------------------------------------------------------------------
    def lookup(self, keys, table_ids, score: ScoreArg, n_dev=None):
        value, policy = self._parse_score(score)
        return ext.table_lookup(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, keys, table_ids,
                                value, policy, num_scores=self.num_scores_, n_dev=n_dev)

    def insert(self, keys, table_ids, score: ScoreArg, insert_results=None, score_out=None, skip=None, indices=None,
               n_dev=None):
        value, policy = self._parse_score(score)
        if os.environ.get("DEMB_DETERMINISM_MODE") is not None and skip is None:
            assert self.num_scores_ == 1
            return self._deterministic_insert(keys, table_ids, value, policy)
        return ext.table_insert(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, self.bucket_sizes,
                                keys, table_ids, value, policy, self._ref_counter, insert_results, score_out,
                                num_scores=self.num_scores_, skip=skip, indices=indices, n_dev=n_dev)

    def insert_and_evict(self, keys, table_ids, score: ScoreArg, insert_results=None, score_out=None):
        value, policy = self._parse_score(score)
        if os.environ.get("DEMB_DETERMINISM_MODE") is not None:
            assert self.num_scores_ == 1
            return self._deterministic_insert_and_evict(keys, table_ids, value, policy)
        idx, nev, ek, ei, es, et = ext.table_insert_and_evict(
            self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, self.bucket_sizes, keys, table_ids,
            value, policy, self._ref_counter, insert_results, score_out, num_scores=self.num_scores_)
        h = int(nev.cpu().item())  # the reference syncs here too (scored_hashtable.py:658)
        return idx, h, ek[:h], ei[:h], es[:h], et[:h]

    def erase(self, keys, table_ids) -> None:
        ext.table_erase(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, self.bucket_sizes, keys,
                        table_ids, num_scores=self.num_scores_)

    def increment_counter(self, slot_indices, table_ids, n_dev=None) -> None:
        ext.table_update_counter_with_layout(self._ref_counter, slot_indices, 1, self.table_bucket_offsets_,
                                             self.bucket_capacity_, self.capacity_, self.num_tables_, table_ids=table_ids,
                                             n_dev=n_dev, **self._ovf())

    def decrement_counter(self, slot_indices, table_ids, n_dev=None) -> None:
        ext.table_update_counter_with_layout(self._ref_counter, slot_indices, -1, self.table_bucket_offsets_,
                                             self.bucket_capacity_, self.capacity_, self.num_tables_, table_ids=table_ids,
                                             n_dev=n_dev, **self._ovf())

    # -- bookkeeping -------------------------------------------------------------------------
    def capacity(self, table_id: Optional[int] = None) -> int:
        """main + overflow (scored_hashtable.py:1379-1391)"""
        if table_id is not None:
            return self.per_table_capacity_[table_id] + self.overflow_bucket_capacity_
        return self.capacity_ + self.overflow_bucket_capacity_ * self.overflow_num_buckets_

    def main_capacity(self, table_id: Optional[int] = None) -> int:
        return self.capacity_ if table_id is None else self.per_table_capacity_[table_id]

    def size(self, table_id: Optional[int] = None):
        if table_id is not None:
            b0 = int(self.table_bucket_offsets_cpu_[table_id])
            b1 = int(self.table_bucket_offsets_cpu_[table_id + 1])
            s = self.bucket_sizes[b0:b1].sum()
            return s + self.overflow_bucket_sizes[table_id] if self.enable_overflow_ else s
        s = self.bucket_sizes.sum()
        return s + self.overflow_bucket_sizes.sum() if self.enable_overflow_ else s

    def load_factor(self) -> float:
        return self.bucket_sizes.sum() / self.capacity_

    def memory_usage(self, mem_type=None) -> int:
        return self.storage_bytes_ + self.bucket_sizes.numel() * self.bucket_sizes.element_size()

    def bucketize_keys(self, keys, table_ids):
        return ext.bucketize_keys(keys, table_ids, self.table_bucket_offsets_, self.num_buckets_, self.bucket_capacity_)

    # -- deterministic mode (scored_hashtable.py:1451-1757) -----------------------------------
    def _waves(self, keys, table_ids, score_value):
        bkt_keys, offsets, inverse = self.bucketize_keys(keys, table_ids)
        lengths = offsets[1:] - offsets[:-1]
        max_len = int(lengths.max().item()) if lengths.numel() else 0
        for w in range(max_len):
            sel = offsets[:-1][lengths > w] + w
            src = inverse[sel]
            yield (bkt_keys[sel].contiguous(), table_ids[src].contiguous(),
                   None if score_value is None else score_value.view(torch.int64)[src].view(score_value.dtype).contiguous())

    def _deterministic_insert(self, keys, table_ids, score_value, policy):
        if keys.numel() == 0:
            return torch.empty(0, dtype=torch.int64, device=keys.device)
        for vk, vt, vs in self._waves(keys, table_ids, score_value):
            ext.table_insert(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, self.bucket_sizes,
                             vk, vt, vs, policy, self._ref_counter)
        _, _, indices = ext.table_lookup(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, keys,
                                         table_ids, None, ScorePolicy.CONST)
        return indices

    def _deterministic_insert_and_evict(self, keys, table_ids, score_value, policy):
        n = keys.numel()
        dev = keys.device
        if n == 0:
            e = torch.empty(0, dtype=torch.int64, device=dev)
            return e, 0, torch.empty_like(keys[:0]), e.clone(), e.clone(), e.clone()
        acc = [[], [], [], []]
        for vk, vt, vs in self._waves(keys, table_ids, score_value):
            _, nev, ek, ei, es, et = ext.table_insert_and_evict(
                self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, self.bucket_sizes, vk, vt, vs,
                policy, self._ref_counter)
            h = int(nev.cpu().item())
            if h:
                for lst, a in zip(acc, (ek, ei, es, et)):
                    lst.append(a[:h])
        _, _, indices = ext.table_lookup(self.table_storage_, self.table_bucket_offsets_, self.bucket_capacity_, keys,
                                         table_ids, None, ScorePolicy.CONST)
        cat = [torch.cat(x) if x else torch.empty(0, dtype=(keys.dtype if i == 0 else torch.int64), device=dev)
               for i, x in enumerate(acc)]
        return indices, cat[0].numel(), cat[0], cat[1], cat[2], cat[3]


def get_scored_table(capacity: List[int], bucket_capacity: Optional[int] = None, key_type=torch.int64,
                     score_specs: List[ScoreSpec] = None, device=None, probing_type=ProbingType.LINEAR,
                     reduction_type=ReductionType.LINEAR, bucket_load_factor=0.5, enable_overflow: bool = False):
    if score_specs is None:
        score_specs = [ScoreSpec(name="timestamp", policy=ScorePolicy.GLOBAL_TIMER)]
    if probing_type == ProbingType.LINEAR and reduction_type == ReductionType.LINEAR:
        return LinearBucketTable(capacity, score_specs, key_type=key_type, bucket_capacity=bucket_capacity, device=device,
                                 enable_overflow=enable_overflow)
    raise NotImplementedError
