"""DynamicEmbeddingBagCollectionSharder / ShardedDynamicEmbeddingBagCollection: pooled dynamic embedding tables behind
TorchRec's EmbeddingBagCollection interface (reference shard/embeddingbag.py:41-101, planner/rw_sharding.py:161-261).

forward(KeyedJaggedTensor) -> KeyedTensor [B, sum of dims].  Every rank pools PARTIAL SUMS of the global batch from its
shard; the output dist adds them (all-to-all of the W blocks + one local add kernel).  MEAN pooling is applied AFTER the
exchange -- divide by the length of the whole bag -- as TorchRec does after its reduce-scatter; a mean taken per shard
would divide by the shard-local length."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
from torch import nn

from .._torchrec import (EmbeddingBagCollection, EmbeddingBagCollectionSharder, EmbeddingComputeKernel, KeyedJaggedTensor,
                         KeyedTensor, NoWait, PoolingType, ShardedModule, ShardingEnv)
from ..batched_dynamicemb_compute_kernel import BatchedDynamicEmbeddingBag
from ..dynamicemb_config import DynamicEmbKernel
from ..input_dist import HipOps
from ..sharded import RowWiseShardedLookup, _ModuleLocal
from .common import DistInput, _LocalLookup, _OutputDist, combined_optimizer, feature_order, group_tables


class DynamicEmbeddingBagCollectionContext:
    def __init__(self) -> None:
        self.dist_input: Optional[DistInput] = None

    def record_stream(self, stream) -> None:
        pass


class ShardedDynamicEmbeddingBagCollection(ShardedModule):
    supported_compute_kernels: List[str] = [k.value for k in EmbeddingComputeKernel] + [DynamicEmbKernel]

    def __init__(self, module: EmbeddingBagCollection, table_name_to_parameter_sharding: Dict[str, Any], env: ShardingEnv,
                 fused_params: Optional[Dict[str, Any]] = None, device: Optional[torch.device] = None,
                 qcomm_codecs_registry: Optional[Dict[str, Any]] = None, module_fqn: Optional[str] = None, ops=None) -> None:
        super().__init__()
        self._env, self._device = env, device
        self._ops = ops or HipOps()
        configs = list(module.embedding_bag_configs())
        groups = group_tables(configs, table_name_to_parameter_sharding, fused_params, False,
                              "ShardedDynamicEmbeddingBagCollection")
        self._groups = groups
        self._feature_names: List[str] = [f for g in groups for f in g.feature_names()]
        self._feature_splits = [len(g.feature_names()) for g in groups]
        self._length_per_key = [t.embedding_dim for g in groups for t in g.embedding_tables for _ in t.feature_names]
        self._kernels = nn.ModuleList()
        self._lookups: List[RowWiseShardedLookup] = []
        self._mean: List[bool] = []
        self._dim_of_col: List[Optional[torch.Tensor]] = []
        for g in groups:
            mean = getattr(g.pooling, "name", str(g.pooling)) == "MEAN"
            self._mean.append(mean)
            if mean:      # the shards pool with SUM; the mean is taken after the output dist
                g.pooling = PoolingType.SUM
            fp = dict(g.fused_params)
            fp["output_dtype"] = torch.float32     # the shards' partial sums leave the local lookup in fp32; on the fabric they
                                                   # travel in fp32 too unless fused_params["wire_dtype"] opts into bf16
            out_dtype = (fused_params or {}).get("output_dtype", torch.float32)
            if not isinstance(out_dtype, torch.dtype):
                out_dtype = out_dtype.as_dtype() if hasattr(out_dtype, "as_dtype") else torch.float32
            g.fused_params = fp
            k = BatchedDynamicEmbeddingBag(g, env.process_group, device)
            self._kernels.append(k)
            hash_sizes = [t.num_embeddings for t in g.embedding_tables for _ in t.feature_names]
            dist_types = [t.fused_params.get("dist_type", "roundrobin") for t in g.embedding_tables for _ in t.feature_names]
            self._lookups.append(RowWiseShardedLookup(_ModuleLocal(k.emb_module), len(hash_sizes), hash_sizes, pooled=True,
                                                      pg=env.process_group, device=device, out_dtype=out_dtype,
                                                      dist_type_per_feature=dist_types, ops=self._ops,
                                                      wire_dtype=(fused_params or {}).get("wire_dtype", None)))
            dims = [t.embedding_dim for t in g.embedding_tables for _ in t.feature_names]
            self._dim_of_col.append(torch.repeat_interleave(torch.arange(len(dims), device=device),
                                                            torch.tensor(dims, device=device)) if mean else None)
        self._order: Optional[List[int]] = None
        self._order_keys: Optional[List[str]] = None
        self._anchor = torch.zeros(1, device=device, requires_grad=True)   # a leaf that gives the autograd nodes an input

    def create_context(self) -> DynamicEmbeddingBagCollectionContext:
        return DynamicEmbeddingBagCollectionContext()

    @property
    def unsharded_module_type(self):
        return EmbeddingBagCollection

    def _ordered(self, features: KeyedJaggedTensor) -> KeyedJaggedTensor:
        keys = list(features.keys())
        if self._order_keys != keys:
            self._order, self._order_keys = feature_order(keys, self._feature_names), keys
        return features if self._order is None else features.permute(self._order)

    def input_dist(self, ctx, features: KeyedJaggedTensor):
        with torch.no_grad():
            features = self._ordered(features)
            B = features.stride()
            sks, lens = [], []
            for gi, part in enumerate(features.split(self._feature_splits)):
                values = part.values().contiguous()
                if values.dtype != torch.int64:
                    values = values.long()
                lens.append(part.lengths())
                sks.append(self._lookups[gi].dist_input(values, part.offsets().to(torch.int64)))
            ctx.dist_input = DistInput(sks, [None] * len(sks), lens, B)
        return NoWait(NoWait(ctx.dist_input))

    def compute(self, ctx, dist_input: DistInput) -> List[torch.Tensor]:
        train = self.training and torch.is_grad_enabled()
        return [_LocalLookup.apply(self._anchor, lk, sk, train) if train else lk.lookup(sk, False)[0]
                for lk, sk in zip(self._lookups, dist_input.sharded_keys)]

    def output_dist(self, ctx, output: List[torch.Tensor]):
        di: DistInput = ctx.dist_input
        blocks = []
        for gi, (lk, sk, out_local) in enumerate(zip(self._lookups, di.sharded_keys, output)):
            pooled = _OutputDist.apply(out_local, lk, sk) if out_local.requires_grad else lk.dist_output(sk, out_local)
            if self._mean[gi]:
                # [B, F_g] bag lengths of MY samples (the whole bag, whatever shard its keys went to) -> per column
                L = di.lengths[gi].view(-1, di.batch_size).t().clamp(min=1).to(pooled.dtype)
                pooled = pooled / L[:, self._dim_of_col[gi]]
            blocks.append(pooled)
        values = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)
        return NoWait(KeyedTensor(keys=self._feature_names, length_per_key=self._length_per_key, values=values))

    def prefetch(self, dist_input, forward_stream=None, ctx=None) -> None:
        """ShardedModule.prefetch of TorchRec's prefetch pipeline (examples/commons/pipeline/train_pipeline.py:663-692 ->
        lookup.prefetch -> emb_module.prefetch, corelib/dynamicemb/dynamicemb/batched_dynamicemb_tables.py:1090-1137): the
        index stage (dedup, find, insert + first-touch init, pin) of this rank's shard for a batch whose input dist is done,
        run on the CURRENT stream -- the pipeline's prefetch stream -- while earlier batches still compute on
        `forward_stream`.  compute() of the same batch then only gathers (the module consumes its prefetch states in FIFO
        order)."""
        di = dist_input
        while not isinstance(di, DistInput) and hasattr(di, "wait"):
            di = di.wait()
        if not (self.training and torch.is_grad_enabled()):
            return
        for lk, sk in zip(self._lookups, di.sharded_keys):
            lk.local.module.prefetch(sk.values, sk.offsets, forward_stream)

    def compute_and_output_dist(self, ctx, input: DistInput):
        return self.output_dist(ctx, self.compute(ctx, input))

    def forward(self, features: KeyedJaggedTensor) -> KeyedTensor:
        ctx = self.create_context()
        di = self.input_dist(ctx, features).wait().wait()
        return self.compute_and_output_dist(ctx, di).wait()

    @property
    def fused_optimizer(self):
        return combined_optimizer(self._kernels)

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        for k in self._kernels:
            yield from k.named_parameters(f"{prefix}.embedding_bags" if prefix else "embedding_bags", recurse, remove_duplicate)

    def state_dict(self, destination=None, prefix: str = "", keep_vars: bool = False):
        for k in self._kernels:
            destination = k.state_dict(destination, prefix + "embedding_bags.", keep_vars)
        return destination

    def dynamic_embedding_modules(self):
        return [k.emb_module for k in self._kernels]


class DynamicEmbeddingBagCollectionSharder(EmbeddingBagCollectionSharder):
    """Drop-in for TorchRec's EmbeddingBagCollectionSharder whose `shard` builds the sharded module of dynamic tables."""

    def shard(self, module: EmbeddingBagCollection, params: Dict[str, Any], env: ShardingEnv,
              device: Optional[torch.device] = None, module_fqn: Optional[str] = None) -> ShardedDynamicEmbeddingBagCollection:
        return ShardedDynamicEmbeddingBagCollection(module=module, table_name_to_parameter_sharding=params, env=env,
                                                    fused_params=self.fused_params, device=device,
                                                    qcomm_codecs_registry=self.qcomm_codecs_registry, module_fqn=module_fqn)
