"""DynamicEmbeddingCollectionSharder / ShardedDynamicEmbeddingCollection: sequence (un-pooled) dynamic embedding tables
behind TorchRec's EmbeddingCollection interface (reference shard/embedding.py:78-394).

forward(KeyedJaggedTensor) -> Dict[feature name, JaggedTensor]; internally input_dist (optional per-table dedup before
the exchange -- `use_index_dedup`, reference :183-275 -- then bucketize + all-to-all of keys), compute (the fused local
lookup), output_dist (all-to-all of rows back, un-bucketize, expand the dedup)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
from torch import nn

import dynamicemb_extensions as ext

from .._torchrec import (EmbeddingCollection, EmbeddingCollectionSharder, EmbeddingComputeKernel, JaggedTensor,
                         KeyedJaggedTensor, NoWait, ShardedModule, ShardingEnv)
from ..batched_dynamicemb_compute_kernel import BatchedDynamicEmbedding
from ..dynamicemb_config import DynamicEmbKernel, DynamicEmbScoreStrategy, get_eviction_score_strategy
from ..input_dist import HipOps
from ..sharded import RowWiseShardedLookup, _ModuleLocal
from .common import DistInput, _Expand, _LocalLookup, _OutputDist, combined_optimizer, feature_order, group_tables


class DynamicEmbeddingCollectionContext:
    """per-forward state between the three stages (TorchRec's EmbeddingCollectionContext + the LFU frequency counters)"""

    def __init__(self) -> None:
        self.dist_input: Optional[DistInput] = None
        self.frequency_counters: List[torch.Tensor] = []

    def record_stream(self, stream) -> None:   # (the train pipelines call this on contexts)
        pass


class ShardedDynamicEmbeddingCollection(ShardedModule):
    supported_compute_kernels: List[str] = [k.value for k in EmbeddingComputeKernel] + [DynamicEmbKernel]

    def __init__(self, module: EmbeddingCollection, table_name_to_parameter_sharding: Dict[str, Any], env: ShardingEnv,
                 fused_params: Optional[Dict[str, Any]] = None, device: Optional[torch.device] = None,
                 qcomm_codecs_registry: Optional[Dict[str, Any]] = None, use_index_dedup: bool = False,
                 score_strategy=None, has_admit_strategy: bool = False, ops=None) -> None:
        super().__init__()
        self._env, self._device = env, device
        self._use_index_dedup = use_index_dedup
        self._is_lfu_enabled = (score_strategy is not None
                                and get_eviction_score_strategy(score_strategy) == DynamicEmbScoreStrategy.LFU)
        self._has_admit_strategy = has_admit_strategy
        self._ops = ops or HipOps()
        groups = group_tables(list(module.embedding_configs()), table_name_to_parameter_sharding, fused_params, True,
                              "ShardedDynamicEmbeddingCollection")
        self._groups = groups
        self._feature_names: List[str] = [f for g in groups for f in g.feature_names()]
        self._feature_splits = [len(g.feature_names()) for g in groups]
        self._dims = {f: t.embedding_dim for g in groups for t in g.embedding_tables for f in t.feature_names}
        self._kernels = nn.ModuleList()
        self._lookups: List[RowWiseShardedLookup] = []
        self._table_offsets: List[torch.Tensor] = []     # per group: first feature of every table (+ end)
        for g in groups:
            k = BatchedDynamicEmbedding(g, env.process_group, device)
            self._kernels.append(k)
            hash_sizes = [t.num_embeddings for t in g.embedding_tables for _ in t.feature_names]
            dist_types = [t.fused_params.get("dist_type", "roundrobin") for t in g.embedding_tables for _ in t.feature_names]
            self._lookups.append(RowWiseShardedLookup(_ModuleLocal(k.emb_module), len(hash_sizes), hash_sizes, pooled=False,
                                                      pg=env.process_group, device=device,
                                                      out_dtype=k.emb_module.output_dtype,
                                                      dist_type_per_feature=dist_types, ops=self._ops))
            tof, n = [0], 0
            for t in g.embedding_tables:
                n += t.num_features()
                tof.append(n)
            self._table_offsets.append(torch.tensor(tof, dtype=torch.int64, device=device))
        self._order: Optional[List[int]] = None
        self._order_keys: Optional[List[str]] = None
        self._anchor = torch.zeros(1, device=device, requires_grad=True)   # a leaf that gives the autograd nodes an input

    # ------------------------------------------------------------------------------------------ ShardedModule protocol
    def create_context(self) -> DynamicEmbeddingCollectionContext:
        return DynamicEmbeddingCollectionContext()

    @property
    def unsharded_module_type(self):
        return EmbeddingCollection

    def _ordered(self, features: KeyedJaggedTensor) -> KeyedJaggedTensor:
        keys = list(features.keys())
        if self._order_keys != keys:
            self._order, self._order_keys = feature_order(keys, self._feature_names), keys
        return features if self._order is None else features.permute(self._order)

    def input_dist(self, ctx: DynamicEmbeddingCollectionContext, features: KeyedJaggedTensor):
        with torch.no_grad():
            features = self._ordered(features)
            B = features.stride()
            sks, revs, lens = [], [], []
            for gi, part in enumerate(features.split(self._feature_splits)):
                values = part.values().contiguous()
                if values.dtype != torch.int64:
                    values = values.long()
                offsets = part.offsets().to(torch.int64)
                lens.append(part.lengths())
                rev = None
                if self._use_index_dedup and values.numel() > 0:
                    # per-table dedup BEFORE the exchange: every key crosses the fabric once per rank
                    tof = self._table_offsets[gi]
                    T = tof.numel() - 1
                    rng = ext.get_table_range(offsets, tof)
                    want_freq = self._is_lfu_enabled or self._has_admit_strategy
                    n_u, ukeys, rev, uoff, freq = ext.segmented_unique_cuda(
                        values, rng, T, torch.empty(0, dtype=torch.int64, device=values.device) if want_freq else None)
                    total_b = offsets.numel() - 1
                    new_len, new_off = ext.compute_dedup_lengths_cuda(uoff, tof, T, B, total_b)
                    nu = int(n_u.item())      # (the one host read the reference has here too, shard/embedding.py:258)
                    values, offsets = ukeys[:nu].contiguous(), new_off
                    if want_freq:
                        ctx.frequency_counters.append(freq[:nu])
                sks.append(self._lookups[gi].dist_input(values, offsets))
                revs.append(rev)
            ctx.dist_input = DistInput(sks, revs, lens, B)
        return NoWait(NoWait(ctx.dist_input))

    def compute(self, ctx, dist_input: DistInput) -> List[torch.Tensor]:
        train = self.training and torch.is_grad_enabled()
        outs = []
        for lk, sk in zip(self._lookups, dist_input.sharded_keys):
            outs.append(_LocalLookup.apply(self._anchor, lk, sk, train) if train else lk.lookup(sk, False)[0])
        return outs

    def output_dist(self, ctx, output: List[torch.Tensor]):
        di: DistInput = ctx.dist_input
        result: Dict[str, JaggedTensor] = {}
        for gi, (lk, sk, out_local) in enumerate(zip(self._lookups, di.sharded_keys, output)):
            rows = _OutputDist.apply(out_local, lk, sk) if out_local.requires_grad else lk.dist_output(sk, out_local)
            if di.reverse[gi] is not None:
                rows = _Expand.apply(rows, di.reverse[gi], self._ops) if rows.requires_grad else \
                    self._ops.gather_rows(rows, di.reverse[gi])
            lengths = di.lengths[gi]
            B = di.batch_size
            per_feature = lengths.view(-1, B).sum(1).tolist() if lengths.numel() else []
            lo = 0
            for fi, name in enumerate(self._groups[gi].feature_names()):
                hi = lo + int(per_feature[fi]) if per_feature else lo
                result[name] = JaggedTensor(values=rows[lo:hi], lengths=lengths[fi * B:(fi + 1) * B])
                lo = hi
        return NoWait(result)

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

    def forward(self, features: KeyedJaggedTensor) -> Dict[str, JaggedTensor]:
        ctx = self.create_context()
        di = self.input_dist(ctx, features).wait().wait()
        return self.compute_and_output_dist(ctx, di).wait()

    # ------------------------------------------------------------------------------------------ optimizer / state
    @property
    def fused_optimizer(self):
        return combined_optimizer(self._kernels)

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        for k in self._kernels:
            yield from k.named_parameters(f"{prefix}.embeddings" if prefix else "embeddings", recurse, remove_duplicate)

    def state_dict(self, destination=None, prefix: str = "", keep_vars: bool = False):
        for k in self._kernels:
            destination = k.state_dict(destination, prefix + "embeddings.", keep_vars)
        return destination

    def dynamic_embedding_modules(self):
        """the BatchedDynamicEmbeddingTablesV2 of every group (what DynamicEmbDump / DynamicEmbLoad walk)"""
        return [k.emb_module for k in self._kernels]


class DynamicEmbeddingCollectionSharder(EmbeddingCollectionSharder):
    """Drop-in for TorchRec's EmbeddingCollectionSharder (same constructor: fused_params, qcomm_codecs_registry,
    use_index_dedup) whose `shard` builds the sharded module of dynamic tables."""

    def shard(self, module: EmbeddingCollection, params: Dict[str, Any], env: ShardingEnv,
              device: Optional[torch.device] = None, module_fqn: Optional[str] = None) -> ShardedDynamicEmbeddingCollection:
        score_strategy, has_admit = None, False
        for ps in params.values():
            o = getattr(ps, "dynamicemb_options", None)
            if o:
                score_strategy = o.score_strategy
                has_admit = o.admit_strategy is not None
                break
        return ShardedDynamicEmbeddingCollection(module, params, env, self.fused_params, device,
                                                 qcomm_codecs_registry=self.qcomm_codecs_registry,
                                                 use_index_dedup=getattr(self, "_use_index_dedup", False),
                                                 score_strategy=score_strategy, has_admit_strategy=has_admit)
