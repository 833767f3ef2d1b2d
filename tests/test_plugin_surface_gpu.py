"""GPU tests of the TorchRec plugin surface (dynamicemb.shard / planner / get_planner / compute kernels) driven through the
protocol stand-ins of tests/standins/torchrec_standin.py when TorchRec is not installed (`test_which_torchrec_world` records
which world a run was in; `test_real_torchrec_embedding_collection_through_dmp` runs only against the real package): get_planner -> collective_plan -> DistributedModelParallel with the
DynamicEmb sharders -> forward / backward on a 1-rank RCCL group, checked against a directly built
BatchedDynamicEmbeddingTablesV2 with the same options (the first-touch initialiser is counter based: same seed + key ->
same row), and the DynamicEmbDump / DynamicEmbLoad round trip.  Mirrors the shape of the reference's distributed tests
(corelib/dynamicemb/test/unit_tests/test_sequence_embedding.sh / test_pooled_embedding.sh: DMP + known-answer rows)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.fixture(scope="module")
def pg():
    import torch.distributed as dist

    from conftest import rendezvous_file
    dist.init_process_group("nccl", init_method=rendezvous_file(), rank=0, world_size=1, device_id=DEV)
    yield dist.group.WORLD
    dist.destroy_process_group()


def _kjt(rng, keys, B, hi, maxlen=5):
    from dynamicemb._torchrec import KeyedJaggedTensor

    lens = rng.integers(0, maxlen, size=len(keys) * B)
    vals = rng.integers(0, hi, size=int(lens.sum())).astype(np.int64)
    return KeyedJaggedTensor(keys, torch.from_numpy(vals).to(DEV), lengths=torch.from_numpy(lens).to(DEV))


def _build(pg, ebc: bool, dedup: bool = False, lr: float = 0.5):
    import dynamicemb as de
    from dynamicemb._torchrec import (DistributedModelParallel, EmbeddingBagCollection, EmbeddingBagConfig, EmbeddingCollection,
                                      EmbeddingConfig, PoolingType, ShardingEnv)
    from dynamicemb.get_planner import get_planner
    from dynamicemb.shard import DynamicEmbeddingBagCollectionSharder, DynamicEmbeddingCollectionSharder

    init = de.DynamicEmbInitializerArgs(mode=de.DynamicEmbInitializerMode.UNIFORM, lower=-0.5, upper=0.5)
    if ebc:
        cfgs = [EmbeddingBagConfig(num_embeddings=5000, embedding_dim=16, name="a", feature_names=["fa0", "fa1"], pooling=PoolingType.SUM),
                EmbeddingBagConfig(num_embeddings=3000, embedding_dim=32, name="b", feature_names=["fb"], pooling=PoolingType.SUM),
                EmbeddingBagConfig(num_embeddings=2000, embedding_dim=8, name="m", feature_names=["fm"], pooling=PoolingType.MEAN)]
        coll = EmbeddingBagCollection(cfgs, device=torch.device("meta"))
    else:
        cfgs = [EmbeddingConfig(num_embeddings=5000, embedding_dim=16, name="a", feature_names=["fa0", "fa1"]),
                EmbeddingConfig(num_embeddings=3000, embedding_dim=16, name="b", feature_names=["fb"])]
        coll = EmbeddingCollection(cfgs, device=torch.device("meta"))
    opts = {c.name: de.DynamicEmbTableOptions(initializer_args=init, score_strategy=de.DynamicEmbScoreStrategy.STEP,
                                              dist_type="hash_roundrobin") for c in cfgs}

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.sparse = coll

        def forward(self, kjt):
            return self.sparse(kjt)

    model = Model()
    planner = get_planner(cfgs, set(), opts, DEV)
    fused = {"optimizer": de.EmbOptimType.SGD, "learning_rate": lr}
    sharders = [DynamicEmbeddingBagCollectionSharder(fused_params=fused),
                DynamicEmbeddingCollectionSharder(fused_params=fused, use_index_dedup=dedup)]
    plan = planner.collective_plan(model, sharders, pg)
    dmp = DistributedModelParallel(module=model, env=ShardingEnv.from_process_group(pg), device=DEV, sharders=sharders, plan=plan,
                                   init_data_parallel=False)
    return dmp, cfgs, opts


def _twin(cfgs, opts, names, pooling, lr):
    """the same tables as ONE directly constructed module (what a single-GPU user of the reference builds)"""
    import dynamicemb as de
    from dynamicemb.batched_dynamicemb_tables import BatchedDynamicEmbeddingTablesV2

    sel = [c for c in cfgs if c.name in names]
    fmap = [i for i, c in enumerate(sel) for _ in c.feature_names]
    return BatchedDynamicEmbeddingTablesV2([opts[c.name] for c in sel], table_names=[c.name for c in sel], feature_table_map=fmap,
                                           pooling_mode=pooling, optimizer=de.EmbOptimType.SGD, learning_rate=lr, device=DEV)


@pytest.mark.parametrize("dedup", [False, True])
def test_embedding_collection_through_sharder_and_dmp(pg, dedup):
    import dynamicemb as de
    from dynamicemb.shard import ShardedDynamicEmbeddingCollection

    dmp, cfgs, opts = _build(pg, ebc=False, dedup=dedup)
    sharded = dmp.module.sparse
    assert isinstance(sharded, ShardedDynamicEmbeddingCollection)
    twin = _twin(cfgs, opts, {"a", "b"}, de.DynamicEmbPoolingMode.NONE, 0.5)
    rng = np.random.default_rng(0)
    dmp.train(); twin.train()
    keys = ["fb", "fa0", "fa1"]          # not the table-major order: the module permutes
    for it in range(3):
        kjt = _kjt(rng, keys, 17, 900)
        out = dmp(kjt)
        assert set(out) == set(keys)
        ordered = kjt.permute([1, 2, 0])
        ref = twin(ordered.values(), ordered.offsets())
        got = torch.cat([out[k].values() for k in ("fa0", "fa1", "fb")])
        assert torch.equal(got, ref)
        for k in keys:
            assert torch.equal(out[k].lengths(), kjt.to_dict()[k].lengths())
        g = torch.rand_like(ref) + 0.1
        got.backward(g)
        ref.backward(g)
    # rows after three SGD steps agree (different summation order of the duplicates' gradients: fp32 rounding)
    for name in ("a", "b"):
        k1, v1 = twin.export_keys_values(name, DEV)
        mod = [m for m in sharded.dynamic_embedding_modules() if name in m._table_names][0]
        k2, v2 = mod.export_keys_values(name, DEV)
        o1, o2 = torch.argsort(k1), torch.argsort(k2)
        assert torch.equal(k1[o1], k2[o2])
        torch.testing.assert_close(v1[o1], v2[o2], rtol=1e-5, atol=1e-6)
    # eval: unknown keys give zero rows, nothing is inserted
    dmp.eval()
    n0 = int(sum(m.size() for m in sharded.dynamic_embedding_modules()))
    with torch.no_grad():
        out = dmp(_kjt(rng, keys, 5, 10 ** 9))
    assert all(bool((out[k].values() == 0).all()) for k in keys)
    assert int(sum(m.size() for m in sharded.dynamic_embedding_modules())) == n0


def test_embedding_bag_collection_with_mean_group_and_optimizer_surface(pg):
    import dynamicemb as de
    from dynamicemb.shard import ShardedDynamicEmbeddingBagCollection

    dmp, cfgs, opts = _build(pg, ebc=True, lr=0.25)
    sharded = dmp.module.sparse
    assert isinstance(sharded, ShardedDynamicEmbeddingBagCollection)
    t_sum = _twin(cfgs, opts, {"a", "b"}, de.DynamicEmbPoolingMode.SUM, 0.25)
    t_mean = _twin(cfgs, opts, {"m"}, de.DynamicEmbPoolingMode.MEAN, 0.25)
    rng = np.random.default_rng(1)
    dmp.train(); t_sum.train(); t_mean.train()
    keys = ["fa0", "fa1", "fb", "fm"]
    for it in range(3):
        kjt = _kjt(rng, keys, 9, 700)
        kt = dmp(kjt)
        assert kt.keys() == keys and kt.length_per_key() == [16, 16, 32, 8]
        parts = kjt.split([3, 1])
        ref = torch.cat([t_sum(parts[0].values(), parts[0].offsets()), t_mean(parts[1].values(), parts[1].offsets())], dim=1)
        torch.testing.assert_close(kt.values(), ref, rtol=1e-6, atol=1e-6)
        g = torch.rand_like(ref) + 0.1
        kt.values().backward(g)
        ref.backward(g)
    for name, tw in (("a", t_sum), ("b", t_sum), ("m", t_mean)):
        k1, v1 = tw.export_keys_values(name, DEV)
        mod = [m for m in sharded.dynamic_embedding_modules() if name in m._table_names][0]
        k2, v2 = mod.export_keys_values(name, DEV)
        o1, o2 = torch.argsort(k1), torch.argsort(k2)
        assert torch.equal(k1[o1], k2[o2])
        torch.testing.assert_close(v1[o1], v2[o2], rtol=1e-5, atol=1e-6)
    # optimizer surface: placeholder parameters marked in-backward, learning rate forwarded by the fused optimizer
    params = dict(dmp.named_parameters())
    assert len(params) == 3 and all(p.device.type == "meta" and hasattr(p, "_in_backward_optimizers") for p in params.values())
    opt = dmp.fused_optimizer
    for grp in opt.param_groups:
        grp["lr"] = 0.125
    opt.step()
    assert all(m.learning_rate == 0.125 for m in sharded.dynamic_embedding_modules())


def test_dump_and_load_through_the_model(pg, tmp_path):
    import dynamicemb as de

    src, cfgs, opts = _build(pg, ebc=False)
    rng = np.random.default_rng(2)
    src.train()
    kjt = _kjt(rng, ["fa0", "fa1", "fb"], 40, 2000)
    out = src(kjt)
    torch.cat([out[k].values() for k in out]).sum().backward()
    de.DynamicEmbDump(str(tmp_path), src, optim=True, pg=pg)
    assert sorted(os.listdir(tmp_path)) == ["model.sparse"]
    assert any(f.startswith("a_emb_keys.rank_0.world_size_1") for f in os.listdir(tmp_path / "model.sparse"))
    with pytest.raises(FileExistsError):
        de.DynamicEmbDump(str(tmp_path), src, pg=pg)
    dst, _, _ = _build(pg, ebc=False)
    de.DynamicEmbLoad(str(tmp_path), dst, optim=True, pg=pg)
    src.eval(); dst.eval()
    with torch.no_grad():
        a, b = src(kjt), dst(kjt)
    for k in a:
        assert torch.equal(a[k].values(), b[k].values())


def test_static_table_in_a_dynamic_collection_is_rejected(pg):
    import dynamicemb as de
    from dynamicemb._torchrec import EmbeddingCollection, EmbeddingConfig, ShardingEnv
    from dynamicemb.get_planner import get_planner
    from dynamicemb.shard import DynamicEmbeddingCollectionSharder

    cfgs = [EmbeddingConfig(num_embeddings=100, embedding_dim=8, name="dyn", feature_names=["f0"]),
            EmbeddingConfig(num_embeddings=100, embedding_dim=8, name="static", feature_names=["f1"])]
    coll = EmbeddingCollection(cfgs, device=torch.device("meta"))
    planner = get_planner(cfgs, set(), {"dyn": de.DynamicEmbTableOptions()}, DEV)
    sh = DynamicEmbeddingCollectionSharder()
    holder = torch.nn.Module()
    holder.ec = coll
    plan = planner.collective_plan(holder, [sh], pg)
    with pytest.raises(NotImplementedError, match="dynamic tables only"):
        sh.shard(coll, plan.plan["ec"], ShardingEnv.from_process_group(pg), DEV)


@pytest.mark.parametrize("ebc", [False, True])
def test_prefetch_hook_of_the_sharded_modules(pg, ebc):
    """ShardedModule.prefetch as TorchRec's prefetch pipeline calls it (examples/commons/pipeline/utils.py:1663-1667:
    keywords ctx / dist_input / forward_stream, on the pipeline's prefetch stream, between input_dist and
    compute_and_output_dist): the index stage of the batch runs ahead on a side stream, compute() then only gathers.  Three
    training steps with the hook against three without it on identically built models: same outputs, same rows."""
    torch.manual_seed(0)
    dmp_a, cfgs, _ = _build(pg, ebc=ebc)
    dmp_b, _, _ = _build(pg, ebc=ebc)
    sa, sb = dmp_a.module.sparse, dmp_b.module.sparse
    keys = ["fa0", "fa1", "fb", "fm"] if ebc else ["fa0", "fa1", "fb"]
    rng = np.random.default_rng(3)
    dmp_a.train(); dmp_b.train()
    side = torch.cuda.Stream()
    for it in range(3):
        kjt = _kjt(rng, keys, 19, 700)
        # (a) the pipeline's sequence: input dist, prefetch on the prefetch stream, compute + output dist on the default one
        ctx = sa.create_context()
        di = sa.input_dist(ctx, kjt).wait().wait()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sa.prefetch(ctx=ctx, dist_input=di, forward_stream=torch.cuda.current_stream())
        torch.cuda.current_stream().wait_stream(side)
        out_a = sa.compute_and_output_dist(ctx, di).wait()
        # (b) plain forward
        out_b = sb(kjt)
        if ebc:
            va, vb = out_a.values(), out_b.values()
        else:
            va = torch.cat([out_a[k].values() for k in keys]); vb = torch.cat([out_b[k].values() for k in keys])
        assert torch.equal(va, vb)
        g = torch.rand_like(va) + 0.1
        va.backward(g); vb.backward(g)
    for ma, mb in zip(sa.dynamic_embedding_modules(), sb.dynamic_embedding_modules()):
        assert not ma._prefetch_states          # every prefetched state was consumed by its forward
        for name in ma._table_names:
            ka, ra = ma.export_keys_values(name, DEV)
            kb, rb = mb.export_keys_values(name, DEV)
            oa, ob = torch.argsort(ka), torch.argsort(kb)
            assert torch.equal(ka[oa], kb[ob])
            torch.testing.assert_close(ra[oa], rb[ob], rtol=1e-6, atol=1e-6)


def test_which_torchrec_world():
    """Records in the test log whether this run exercised the plugin surface against the REAL TorchRec or against the
    protocol stand-ins (tests/standins/torchrec_standin.py) -- so that a green plugin-surface suite cannot be read as
    'drops into TorchRec' when TorchRec was never imported."""
    from dynamicemb import _torchrec

    world = "real torchrec" if _torchrec.HAVE_TORCHREC else "stand-ins (torchrec not installed)"
    print(f"plugin surface tested against: {world}")
    if not _torchrec.HAVE_TORCHREC:
        import torchrec_standin   # noqa: F401  (the module the product bound to must be the one under tests/)

        assert os.path.dirname(os.path.abspath(torchrec_standin.__file__)).endswith(os.path.join("tests", "standins"))


def test_real_torchrec_embedding_collection_through_dmp(pg):
    """The reference's own wiring (examples/commons/distributed/sharding.py:156-267): an `EmbeddingCollection` on the meta
    device pushed through TorchRec's REAL `DistributedModelParallel` with `DynamicEmbeddingCollectionSharder`, the plan from
    `get_planner(...).collective_plan`, one forward / backward, rows compared with a directly built module.  Needs the real
    package: skipped -- explicitly, so that the GPU test record says so -- where TorchRec is not installed."""
    from dynamicemb import _torchrec

    if not _torchrec.HAVE_TORCHREC:
        pytest.skip("torchrec not installed: the plugin surface ran against tests/standins/torchrec_standin.py only")
    import dynamicemb as de
    from dynamicemb.shard import ShardedDynamicEmbeddingCollection

    dmp, cfgs, opts = _build(pg, ebc=False, dedup=False)
    sharded = dmp.module.sparse
    assert isinstance(sharded, ShardedDynamicEmbeddingCollection)
    twin = _twin(cfgs, opts, {"a", "b"}, de.DynamicEmbPoolingMode.NONE, 0.5)
    rng = np.random.default_rng(0)
    dmp.train(); twin.train()
    keys = ["fb", "fa0", "fa1"]
    kjt = _kjt(rng, keys, 17, 900)
    out = dmp(kjt)
    out = out.wait() if hasattr(out, "wait") else out
    ordered = kjt.permute([1, 2, 0])
    ref = twin(ordered.values(), ordered.offsets())
    got = torch.cat([out[k].values() for k in ("fa0", "fa1", "fb")])
    assert torch.equal(got, ref)
    g = torch.rand_like(ref) + 0.1
    got.backward(g)
    ref.backward(g)
