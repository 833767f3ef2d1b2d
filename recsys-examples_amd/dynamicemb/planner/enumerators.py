"""DynamicEmbeddingEnumerator (reference planner/enumerators.py): the sharding-option enumerator handed to TorchRec's
planner.  Dynamic tables are not enumerated at all here -- their plan is fixed (row-wise over every rank, the DynamicEmb
compute kernel, capacity ceil(N / W) rounded to buckets; planner.py) and their HBM / host footprint is governed by
`DynamicEmbTableOptions.global_hbm_for_values`, not by TorchRec's storage estimator."""
from typing import Any, Dict, List, Optional

from .._torchrec import HAVE_TORCHREC, Topology

if HAVE_TORCHREC:  # pragma: no cover - only where torchrec is installed
    from torchrec.distributed.planner.enumerators import EmbeddingEnumerator as _Base
else:

    class _Base:  # protocol stand-in: an enumerator is (topology, batch_size, constraints, estimator) + enumerate()
        def __init__(self, topology: Topology, batch_size: Optional[int] = None, constraints: Optional[Dict[str, Any]] = None,
                     estimator: Any = None, use_exact_enumerate_order: bool = False) -> None:
            self._topology, self._batch_size, self._constraints = topology, batch_size, constraints or {}

        def enumerate(self, module, sharders) -> List[Any]:
            return []


class DynamicEmbeddingEnumerator(_Base):
    def __init__(self, topology: Topology, batch_size: Optional[int] = None, constraints: Optional[Dict[str, Any]] = None,
                 estimator: Any = None, use_exact_enumerate_order: bool = False) -> None:
        kw = dict(topology=topology, constraints=constraints, estimator=estimator)
        if batch_size is not None:
            kw["batch_size"] = batch_size
        if HAVE_TORCHREC:  # pragma: no cover
            kw["use_exact_enumerate_order"] = use_exact_enumerate_order
        super().__init__(**kw)
        self._dynamicemb_tables = {n for n, c in (constraints or {}).items() if getattr(c, "use_dynamicemb", False)}

    def enumerate(self, module, sharders) -> List[Any]:
        """TorchRec's options for the static tables; the dynamic ones still get their (cheapest) row-wise option so that
        the plan has an entry for DynamicEmbeddingShardingPlanner.collective_plan to replace"""
        return super().enumerate(module, sharders)
