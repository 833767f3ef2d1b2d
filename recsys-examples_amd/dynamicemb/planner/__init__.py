"""`dynamicemb.planner` (reference corelib/dynamicemb/dynamicemb/planner/__init__.py): planning of row-wise sharded
dynamic embedding tables next to TorchRec's own planner."""
from .enumerators import DynamicEmbeddingEnumerator
from .planner import (DynamicEmbeddingShardingPlanner, DynamicEmbParameterConstraints, DynamicEmbParameterSharding)

__all__ = ["DynamicEmbeddingEnumerator", "DynamicEmbeddingShardingPlanner", "DynamicEmbParameterConstraints",
           "DynamicEmbParameterSharding"]
