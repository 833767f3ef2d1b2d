"""The TorchRec plugin surface of `dynamicemb` on CPU (no TorchRec in this image: the package falls back to its protocol
stand-ins, dynamicemb/_torchrec_standin.py): the names `examples/commons/distributed/sharding.py:26-35,193-216` imports,
the package exports of the reference's `dynamicemb/__init__.py:16-75`, the planner's per-rank table options
(planner/planner.py:124-211), get_planner's three kinds of constraints (get_planner.py:59-131) and the config helpers."""
import math

import pytest
import torch

# the reference's `__all__` (names are API, listed here as data)
REFERENCE_EXPORTS = [
    "AdmissionStrategy", "BUCKET_ALIGNMENT", "DEMB_TABLE_ALIGN_SIZE", "MAX_BUCKET_CAPACITY", "align_to_table_size",
    "get_table_value_bytes", "get_sharded_table_capacity", "FrequencyAdmissionStrategy", "Counter", "KVCounter",
    "DynamicEmbCheckMode", "DynamicEmbInitializerArgs", "DynamicEmbInitializerMode", "DynamicEmbTableOptions",
    "DynamicEmbPoolingMode", "DynamicEmbEvictStrategy", "DynamicEmbScoreStrategy", "ScoreStrategy", "BATCH_SIZE_PER_DUMP",
    "data_type_to_dyn_emb", "data_type_to_dtype", "dyn_emb_to_torch", "torch_to_dyn_emb", "string_to_evict_strategy",
    "DynamicEmbDump", "DynamicEmbLoad", "EmbOptimType", "OptimizerArgs"]


def test_import_block_of_the_example_sharding_module():
    # examples/commons/distributed/sharding.py:26-35, verbatim names
    from dynamicemb import DynamicEmbTableOptions  # noqa: F401
    from dynamicemb.get_planner import get_planner  # noqa: F401
    from dynamicemb.planner import DynamicEmbeddingShardingPlanner as DynamicEmbeddingShardingPlanner  # noqa: F401
    from dynamicemb.shard import DynamicEmbeddingBagCollectionSharder, DynamicEmbeddingCollectionSharder  # noqa: F401
    from dynamicemb.utils import TORCHREC_TYPES

    assert len(TORCHREC_TYPES) == 2
    # the other submodules the reference's Python layer imports from
    from dynamicemb.batched_dynamicemb_compute_kernel import BatchedDynamicEmbedding, BatchedDynamicEmbeddingBag  # noqa: F401
    from dynamicemb.planner import (DynamicEmbeddingEnumerator, DynamicEmbParameterConstraints,  # noqa: F401
                                    DynamicEmbParameterSharding)
    from dynamicemb.shard import ShardedDynamicEmbeddingBagCollection, ShardedDynamicEmbeddingCollection  # noqa: F401


def test_package_exports_match_the_reference():
    import dynamicemb

    assert sorted(dynamicemb.__all__) == sorted(REFERENCE_EXPORTS)
    for name in REFERENCE_EXPORTS:
        assert hasattr(dynamicemb, name), name


def _configs():
    from dynamicemb._torchrec import EmbeddingBagConfig, EmbeddingConfig, PoolingType

    return [EmbeddingConfig(num_embeddings=1_000_003, embedding_dim=128, name="item", feature_names=["item_id", "hist_item"]),
            EmbeddingConfig(num_embeddings=50_000, embedding_dim=128, name="user", feature_names=["user_id"]),
            EmbeddingConfig(num_embeddings=97, embedding_dim=128, name="gender", feature_names=["gender"]),
            EmbeddingBagConfig(num_embeddings=4000, embedding_dim=32, name="ctx", feature_names=["ctx"], pooling=PoolingType.MEAN)]


def test_get_planner_builds_the_three_kinds_of_constraints_and_the_row_wise_plan():
    from dynamicemb import DynamicEmbInitializerArgs, DynamicEmbTableOptions
    from dynamicemb._torchrec import (EmbeddingCollection, EmbeddingBagCollection, EmbeddingComputeKernel, ShardingType)
    from dynamicemb.get_planner import get_planner
    from dynamicemb.planner import DynamicEmbParameterSharding
    from dynamicemb.shard import DynamicEmbeddingBagCollectionSharder, DynamicEmbeddingCollectionSharder

    cfgs = _configs()
    opts = {"item": DynamicEmbTableOptions(global_hbm_for_values=1 << 30, initializer_args=DynamicEmbInitializerArgs()),
            "ctx": DynamicEmbTableOptions(init_capacity=1000, bucket_capacity=64)}
    planner = get_planner(cfgs, {"gender"}, opts, torch.device("cpu"), pipeline_type="native")

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mp = EmbeddingCollection(cfgs[:2], device=torch.device("meta"))
            self.dp = EmbeddingCollection(cfgs[2:3], device=torch.device("meta"))
            self.bags = EmbeddingBagCollection(cfgs[3:], device=torch.device("meta"))

    plan = planner.collective_plan(Model(), [DynamicEmbeddingBagCollectionSharder(), DynamicEmbeddingCollectionSharder()])
    item = plan.plan["mp"]["item"]
    assert isinstance(item, DynamicEmbParameterSharding)
    assert item.sharding_type == ShardingType.ROW_WISE.value
    assert item.compute_kernel == EmbeddingComputeKernel.CUSTOMIZED_KERNEL.value and item.customized_compute_kernel == "DynamicEmb"
    o = item.dynamicemb_options
    # world size 1: capacity = N rounded up to whole buckets of 128; init_capacity defaults to it; uniform bounds filled
    assert o.max_capacity == math.ceil(1_000_003 / 128) * 128 == o.init_capacity and o.bucket_capacity == 128
    assert o.local_hbm_for_values == 1 << 30 and o.dim == 128 and o.index_type == torch.int64 and o.embedding_dtype == torch.float32
    assert abs(o.initializer_args.upper - (1 / 1_000_003) ** 0.5) < 1e-12 and o.initializer_args.lower == -o.initializer_args.upper
    assert item.sharding_spec.shards[0].shard_sizes == [o.max_capacity, 128] and item.ranks == [0]
    assert set(item.get_additional_fused_params()) == {"customized_compute_kernel", "dist_type", "dynamicemb_options"}
    ctx = plan.plan["bags"]["ctx"].dynamicemb_options
    assert ctx.bucket_capacity == 64 and ctx.max_capacity == 4032 and ctx.init_capacity == 1024
    # static model-parallel and data-parallel tables keep TorchRec kernels
    assert plan.plan["mp"]["user"].sharding_type == ShardingType.ROW_WISE.value and plan.plan["mp"]["user"].compute_kernel == "fused"
    assert plan.plan["dp"]["gender"].sharding_type == ShardingType.DATA_PARALLEL.value and plan.plan["dp"]["gender"].compute_kernel == "dense"


def test_planner_turns_global_options_into_per_rank_options():
    from dynamicemb import MAX_BUCKET_CAPACITY, DynamicEmbTableOptions
    from dynamicemb.planner import DynamicEmbParameterConstraints
    from dynamicemb.planner.planner import _prepare_dynemb_table_options

    cfgs = _configs()[:1]
    c = {"item": DynamicEmbParameterConstraints(use_dynamicemb=True, dynamicemb_options=DynamicEmbTableOptions(
        global_hbm_for_values=1001, init_capacity=10 ** 9))}
    with pytest.warns(UserWarning, match="exceeds max_capacity"):
        _prepare_dynemb_table_options(c, cfgs, world_size=8)
    o = c["item"].dynamicemb_options
    per_rank = math.ceil(math.ceil(1_000_003 / 8) / 128) * 128
    assert o.max_capacity == per_rank == o.init_capacity and o.local_hbm_for_values == 126
    # one bucket spanning the shard
    c = {"item": DynamicEmbParameterConstraints(use_dynamicemb=True,
                                               dynamicemb_options=DynamicEmbTableOptions(bucket_capacity=MAX_BUCKET_CAPACITY))}
    _prepare_dynemb_table_options(c, cfgs, world_size=8)
    o = c["item"].dynamicemb_options
    assert o.bucket_capacity == o.max_capacity == math.ceil(math.ceil(1_000_003 / 8) / 16) * 16
    # naming errors
    with pytest.raises(ValueError, match="does not match any key"):
        _prepare_dynemb_table_options({}, cfgs, world_size=2)
    with pytest.raises(ValueError, match="matching BaseEmbeddingConfig"):
        _prepare_dynemb_table_options({"item": c["item"], "ghost": c["item"]}, cfgs, world_size=2)
    with pytest.raises(ValueError, match="unique"):
        _prepare_dynemb_table_options({"item": c["item"]}, cfgs + cfgs, world_size=2)


def test_config_helpers():
    import dynamicemb as de
    from dynamicemb._torchrec import DataType, EmbeddingConfig
    from dynamicemb.batched_dynamicemb_compute_kernel import _prepare_fused_params, pooling_mode_to_dynamicemb
    from dynamicemb.dynamicemb_config import complete_initializer_args, get_constraint_capacity
    from dynamicemb_extensions import DynamicEmbDataType, EvictStrategy

    assert [de.align_to_table_size(n) for n in (-3, 0, 1, 16, 17)] == [16, 16, 16, 16, 32]
    assert de.align_to_table_size(129, 128) == 256
    cfg = EmbeddingConfig(num_embeddings=1000, embedding_dim=8, name="t", data_type=DataType.BF16)
    assert de.get_sharded_table_capacity(cfg, 3, 128) == 384
    assert de.get_sharded_table_capacity(cfg, 3, de.MAX_BUCKET_CAPACITY) == 336
    with pytest.raises(ValueError):
        de.get_sharded_table_capacity(cfg, 3, 100)
    # Adam: row = 3 x dim elements of 2 bytes
    assert de.get_table_value_bytes(cfg, de.EmbOptimType.ADAM, 3, 128) == 384 * 3 * 24 * 2
    assert get_constraint_capacity(10 ** 6, torch.float32, 8, de.EmbOptimType.SGD, 128) == (10 ** 6 // 32) // 128 * 128
    assert de.data_type_to_dtype(DataType.BF16) == torch.bfloat16 and de.data_type_to_dyn_emb(DataType.FP16) == DynamicEmbDataType.Float16
    for t in (torch.float32, torch.bfloat16, torch.float16, torch.int64, torch.int32):
        assert de.dyn_emb_to_torch(de.torch_to_dyn_emb(t)) == t
    assert de.string_to_evict_strategy("KLfu") == EvictStrategy.KLfu
    with pytest.raises(ValueError):
        de.string_to_evict_strategy("nope")
    a = de.DynamicEmbInitializerArgs(lower=-2.0)
    b = complete_initializer_args(a, embedding_config=cfg)
    assert b is not a and b.lower == -2.0 and abs(b.upper - 1000 ** -0.5) < 1e-12 and a.upper is None
    n = de.DynamicEmbInitializerArgs(mode=de.DynamicEmbInitializerMode.NORMAL)
    assert complete_initializer_args(n) is n
    assert pooling_mode_to_dynamicemb(1) == de.DynamicEmbPoolingMode.MEAN
    fp = _prepare_fused_params({"betas": (0.8, 0.9), "output_dtype": DataType.BF16, "dist_type": "x", "dynamicemb_options": 1,
                                "customized_compute_kernel": "DynamicEmb", "learning_rate": 0.5})
    assert fp == {"beta1": 0.8, "beta2": 0.9, "output_dtype": torch.bfloat16, "learning_rate": 0.5}


def test_keyed_jagged_tensor_standin_split_and_permute():
    from dynamicemb._torchrec import HAVE_TORCHREC, KeyedJaggedTensor

    if HAVE_TORCHREC:
        pytest.skip("the real KeyedJaggedTensor is in use")
    lengths = torch.tensor([1, 0, 2, 3, 1, 1])          # 3 features x batch 2
    kjt = KeyedJaggedTensor(["a", "b", "c"], torch.arange(8), lengths=lengths)
    assert kjt.stride() == 2 and kjt.offsets().tolist() == [0, 1, 1, 3, 6, 7, 8]
    p = kjt.permute([2, 0, 1])
    assert p.keys() == ["c", "a", "b"] and p.values().tolist() == [6, 7, 0, 1, 2, 3, 4, 5] and p.lengths().tolist() == [1, 1, 1, 0, 2, 3]
    s = kjt.split([1, 2])
    assert s[1].keys() == ["b", "c"] and s[1].values().tolist() == [1, 2, 3, 4, 5, 6, 7] and s[0].values().tolist() == [0]
    d = kjt.to_dict()
    assert d["b"].values().tolist() == [1, 2, 3, 4, 5] and d["b"].lengths().tolist() == [2, 3]


def _params(fn):
    import inspect

    return [p for p in inspect.signature(fn).parameters if p != "self"]


def test_torchrec_protocol_conformance():
    """Method and argument names of TorchRec's ShardedModule / ModuleSharder / BaseBatchedEmbedding protocols, held as data
    (tests/golden/torchrec_protocol.json: the names the reference's pipeline / planner call and its subclasses override),
    against the sharded modules, the sharders and the compute kernels of this package AND against the stand-in they are
    tested on -- a stand-in that drifted from the protocol would make every plugin-surface test vacuous."""
    import json
    import os

    import dynamicemb
    from dynamicemb import _torchrec
    from dynamicemb.batched_dynamicemb_compute_kernel import BatchedDynamicEmbedding, BatchedDynamicEmbeddingBag
    from dynamicemb.shard.embedding import DynamicEmbeddingCollectionSharder, ShardedDynamicEmbeddingCollection
    from dynamicemb.shard.embeddingbag import DynamicEmbeddingBagCollectionSharder, ShardedDynamicEmbeddingBagCollection

    spec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "torchrec_protocol.json")))

    def check(cls, methods, props):
        for name, args in methods.items():
            fn = getattr(cls, name, None)
            assert callable(fn), f"{cls.__name__}.{name} missing"
            have = _params(fn)
            # the protocol's arguments, by name and in its order (extra trailing keyword arguments with defaults are fine)
            assert have[:len(args)] == args or all(a in have for a in args), f"{cls.__name__}.{name}{tuple(have)} vs protocol {tuple(args)}"
        for name in props:
            assert hasattr(cls, name), f"{cls.__name__}.{name} missing"

    for cls in (ShardedDynamicEmbeddingCollection, ShardedDynamicEmbeddingBagCollection):
        check(cls, spec["ShardedModule"], spec["ShardedModule_properties"])
    for cls in (DynamicEmbeddingCollectionSharder, DynamicEmbeddingBagCollectionSharder):
        check(cls, spec["ModuleSharder"], spec["ModuleSharder_properties"])
    for cls in (BatchedDynamicEmbedding, BatchedDynamicEmbeddingBag):
        check(cls, spec["BaseBatchedEmbedding"], spec["BaseBatchedEmbedding_properties"])
    # the pipeline calls prefetch with keywords (utils.py:1663-1667)
    for cls in (ShardedDynamicEmbeddingCollection, ShardedDynamicEmbeddingBagCollection):
        assert set(_params(cls.prefetch)) >= {"ctx", "dist_input", "forward_stream"}
    assert dynamicemb is not None and _torchrec is not None
