"""One place that resolves the TorchRec names the plugin surface is written against: the real package when it can be
imported (torchrec release/V1.5.0 is what the reference pins, docker/Dockerfile:27).  TorchRec is not installed in the
build image; the TEST SUITE then puts `tests/standins/torchrec_standin.py` (protocol fakes: test infrastructure, not
product) on sys.path and this module binds to it -- `HAVE_TORCHREC` says which world a run was in.  With neither, importing
the plugin surface (sharders, planner, compute kernels, `utils.TORCHREC_TYPES`) fails loudly; the lookup modules, the
extension ops and `hstu` do not need TorchRec and import without it."""
try:  # pragma: no cover - exercised only where torchrec is installed
    import torchrec  # noqa: F401
    from torchrec.distributed.comm import get_local_size
    from torchrec.distributed.embedding import EmbeddingCollectionSharder
    from torchrec.distributed.embedding_types import EmbeddingComputeKernel
    from torchrec.distributed.embeddingbag import EmbeddingBagCollectionSharder
    from torchrec.distributed.model_parallel import DistributedModelParallel
    from torchrec.distributed.planner import ParameterConstraints, Topology
    from torchrec.distributed.planner.storage_reservations import HeuristicalStorageReservation
    from torchrec.distributed.sharding_plan import placement
    from torchrec.distributed.types import (Awaitable, BoundsCheckMode, EmbeddingModuleShardingPlan, EnumerableShardingSpec,
                                            LazyAwaitable, ModuleSharder, NoWait, ParameterSharding, ShardedModule,
                                            ShardingEnv, ShardingPlan, ShardingType, ShardMetadata)
    from torchrec.modules.embedding_configs import (BaseEmbeddingConfig, DataType, EmbeddingBagConfig, EmbeddingConfig,
                                                    PoolingType, data_type_to_dtype)
    from torchrec.modules.embedding_modules import EmbeddingBagCollection, EmbeddingCollection
    from torchrec.optim.fused import EmptyFusedOptimizer, FusedOptimizer, FusedOptimizerModule
    from torchrec.optim.keyed import CombinedOptimizer
    from torchrec.optim.optimizers import in_backward_optimizer_filter
    from torchrec.sparse.jagged_tensor import JaggedTensor, KeyedJaggedTensor, KeyedTensor

    HAVE_TORCHREC = True
except ImportError:
    try:
        from torchrec_standin import (Awaitable, BaseEmbeddingConfig, BoundsCheckMode, CombinedOptimizer, DataType,  # noqa: F401
                                        DistributedModelParallel, EmbeddingBagCollection, EmbeddingBagCollectionSharder,
                                        EmbeddingBagConfig, EmbeddingCollection, EmbeddingCollectionSharder,
                                        EmbeddingComputeKernel, EmbeddingConfig, EmbeddingModuleShardingPlan,
                                        EmptyFusedOptimizer, EnumerableShardingSpec, FusedOptimizer, FusedOptimizerModule,
                                        HeuristicalStorageReservation, JaggedTensor, KeyedJaggedTensor, KeyedTensor,
                                        LazyAwaitable, ModuleSharder, NoWait, ParameterConstraints, ParameterSharding,
                                        PoolingType, ShardedModule, ShardingEnv, ShardingPlan, ShardingType, ShardMetadata,
                                        Topology, data_type_to_dtype, get_local_size, in_backward_optimizer_filter, placement)
    except ImportError as e:
        raise ImportError("TorchRec is not installed (and no test stand-in is on sys.path): the DynamicEmb TorchRec plugin "
                          "surface -- dynamicemb.shard, dynamicemb.planner, get_planner, utils.TORCHREC_TYPES -- needs it") from e

    HAVE_TORCHREC = False
