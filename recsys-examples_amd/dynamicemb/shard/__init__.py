"""`dynamicemb.shard` (reference corelib/dynamicemb/dynamicemb/shard/__init__.py): the TorchRec sharders of dynamic
embedding tables and the sharded modules they build."""
from .embedding import DynamicEmbeddingCollectionSharder, ShardedDynamicEmbeddingCollection
from .embeddingbag import DynamicEmbeddingBagCollectionSharder, ShardedDynamicEmbeddingBagCollection

__all__ = ["ShardedDynamicEmbeddingCollection", "DynamicEmbeddingCollectionSharder",
           "ShardedDynamicEmbeddingBagCollection", "DynamicEmbeddingBagCollectionSharder"]
