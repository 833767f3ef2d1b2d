"""Names of the reference's `dynamicemb.types` that the package exports (types.py:33-116,333-401)."""
from .dynamicemb_config import (BUCKET_ALIGNMENT, DEMB_TABLE_ALIGN_SIZE, EMBEDDING_TYPE, KEY_TYPE,  # noqa: F401
                                MAX_BUCKET_CAPACITY, SCORE_TYPE, DynamicEmbInitializerArgs, DynamicEmbInitializerMode)
from .embedding_admission import AdmissionStrategy, Counter  # noqa: F401

OPT_STATE_TYPE = EMBEDDING_TYPE
COUNTER_TYPE = SCORE_TYPE


import abc as _abc
import enum as _enum


class CopyMode(_enum.Enum):
    """what `Storage.find` copies out per key (reference types.py:133-146): the embedding columns only (eval), or the
    whole row = embedding + optimizer state (training)"""
    EMBEDDING = "embedding"
    VALUE = "value"


class Storage(_abc.ABC):
    """The interface a user-supplied key-value store implements to back a table (`DynamicEmbTableOptions.external_storage`;
    reference types.py:149-288).  The module hands it de-duplicated keys with their table ids and works on the dense value
    buffer it returns: gather / pooling, gradient reduction and the optimizer step run on the device, the store only finds
    and inserts rows.

    find(unique_keys, table_ids, copy_mode, lfu_accumulated_frequency=None) ->
        (num_missing: int, missing_keys, missing_indices, [missing_table_ids,] missing_scores, founds, output_scores, values)
        values: [n, max_emb_dim] (CopyMode.EMBEDDING) or [n, max_value_dim] (CopyMode.VALUE) on the device; rows of missing
        keys are unspecified.  (The reference's own implementations return the 8-tuple with missing_table_ids, its abstract
        class documents the 7-tuple: both are accepted.)
    insert(keys, table_ids, values, scores=None, preserve_existing=False)
    """

    @_abc.abstractmethod
    def find(self, unique_keys, table_ids, copy_mode, lfu_accumulated_frequency=None):
        ...

    @_abc.abstractmethod
    def insert(self, keys, table_ids, values, scores=None, preserve_existing=False) -> None:
        ...

    def dump(self, table_id, meta_file_path, emb_key_path, embedding_file_path, score_file_path, opt_file_path, **kwargs) -> None:
        raise NotImplementedError

    def load(self, table_id, meta_file_path, emb_file_path, embedding_file_path, score_file_path, opt_file_path, **kwargs) -> None:
        raise NotImplementedError

    def export_keys_values(self, device, batch_size: int = 65536, table_id: int = 0):
        raise NotImplementedError

    def size(self) -> int:
        raise NotImplementedError
