"""What the two sharded modules share: grouping the dynamic tables of a collection, the feature order, the three
forward stages over `RowWiseShardedLookup` (input dist / local lookup / output dist) as autograd nodes.

The reference gets the exchange from TorchRec (KJTAllToAll in, SequenceEmbeddingsAllToAll / PooledEmbeddingsReduceScatter
out, hooked in planner/rw_sharding.py:85-261 and shard/embedding.py:78-394).  Here the exchange is this repo's own
(dynamicemb/sharded.py, input_dist.py): bucketize kernel -> RCCL all-to-all of keys -> local fused lookup -> all-to-all of
rows (sequence) / of partial sums + one local add (pooled; a full xGMI mesh carries an all-to-all on 7 links in parallel
where a ring reduce-scatter is bound by one)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist

from .._torchrec import CombinedOptimizer, EmbeddingComputeKernel, ShardingType
from ..batched_dynamicemb_compute_kernel import GroupedTables, ShardedTable
from ..dynamicemb_config import DynamicEmbKernel


def is_dynamic(ps) -> bool:
    ck = getattr(ps, "compute_kernel", None)
    return (ck in (EmbeddingComputeKernel.CUSTOMIZED_KERNEL.value, EmbeddingComputeKernel.CUSTOMIZED_KERNEL)
            and getattr(ps, "customized_compute_kernel", DynamicEmbKernel) == DynamicEmbKernel
            and getattr(ps, "dynamicemb_options", None) is not None)


def group_tables(configs: List[Any], params: Dict[str, Any], fused_params: Optional[Dict[str, Any]], same_dim: bool,
                 what: str) -> List[GroupedTables]:
    """dynamic tables of a collection -> groups that can share one BatchedDynamicEmbeddingTablesV2: equal grouped option
    keys (DynamicEmbTableOptions.__eq__), equal pooling, and for sequence embeddings equal dim"""
    groups: List[GroupedTables] = []
    keys: List[Any] = []
    for cfg in configs:
        ps = params.get(cfg.name)
        if ps is None:
            raise ValueError(f"{what}: table {cfg.name!r} has no entry in the sharding plan")
        if not is_dynamic(ps):
            raise NotImplementedError(
                f"{what}: table {cfg.name!r} is not planned for the DynamicEmb kernel.  A collection sharded by a DynamicEmb "
                "sharder must hold dynamic tables only (static tables belong to a collection of their own, sharded by "
                "TorchRec's sharders -- the layout examples/hstu uses).")
        if ps.sharding_type != ShardingType.ROW_WISE.value:
            raise NotImplementedError(f"{what}: dynamic tables are row-wise sharded, got {ps.sharding_type!r} for {cfg.name!r}")
        o = ps.dynamicemb_options
        tf = dict(fused_params or {})
        tf.update(ps.get_additional_fused_params() if hasattr(ps, "get_additional_fused_params") else
                  {"dynamicemb_options": o, "dist_type": getattr(ps, "dist_type", o.dist_type)})
        pooling = getattr(cfg, "pooling", None)
        t = ShardedTable(name=cfg.name, embedding_dim=cfg.embedding_dim, local_rows=o.max_capacity, local_cols=cfg.embedding_dim,
                         feature_names=list(cfg.feature_names), fused_params=tf, pooling=pooling,
                         num_embeddings=cfg.num_embeddings)
        key = (o, pooling, cfg.embedding_dim if same_dim else None, cfg.data_type)
        for g, k in zip(groups, keys):
            if k[0] == key[0] and k[1:] == key[1:]:
                g.embedding_tables.append(t)
                break
        else:
            groups.append(GroupedTables([t], data_type=cfg.data_type, pooling=pooling, fused_params=dict(fused_params or {})))
            keys.append(key)
    return groups


class _LocalLookup(torch.autograd.Function):
    """stage 2: the fused lookup of this rank's shard on the keys it received; backward = fused reduce + optimizer"""

    @staticmethod
    def forward(ctx, anchor, lookup, sk, train):
        out, lctx = lookup.lookup(sk, train)
        ctx.lookup, ctx.lctx = lookup, lctx
        return out

    @staticmethod
    def backward(ctx, g):
        ctx.lookup.local.backward(ctx.lctx, g.contiguous())
        return None, None, None, None


class _OutputDist(torch.autograd.Function):
    """stage 3: rows / partial sums back to the ranks that asked; backward = the reverse exchange of the gradients"""

    @staticmethod
    def forward(ctx, out_local, lookup, sk):
        ctx.lookup, ctx.sk = lookup, sk
        return lookup.dist_output(sk, out_local)

    @staticmethod
    def backward(ctx, g):
        return ctx.lookup.dist_grads(ctx.sk, g), None, None


class _Expand(torch.autograd.Function):
    """rows of the unique keys -> rows of all keys (pre-communication dedup); backward sums the gradients per unique key"""

    @staticmethod
    def forward(ctx, rows_u, rev, ops):
        ctx.rev, ctx.nu, ctx.ops = rev, rows_u.size(0), ops
        return ops.gather_rows(rows_u, rev)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ug = ctx.ops.reduce_grads(ctx.rev, g, ctx.nu, ctx.rev.numel(), g.size(1), None, None, -1)
        return ug.to(g.dtype), None, None


@dataclass
class DistInput:
    """what input_dist leaves for compute: per group the keys this rank owns (ShardedKeys) and the dedup bookkeeping"""
    sharded_keys: List[Any]
    reverse: List[Optional[torch.Tensor]]
    lengths: List[torch.Tensor]     # per group: lengths [F_g * B] of the ORIGINAL (not de-duplicated) features
    batch_size: int


def feature_order(input_keys: List[str], wanted: List[str]) -> Optional[List[int]]:
    if list(input_keys) == list(wanted):
        return None
    pos = {k: i for i, k in enumerate(input_keys)}
    missing = [k for k in wanted if k not in pos]
    if missing:
        raise KeyError(f"features {missing} are not in the input KeyedJaggedTensor")
    return [pos[k] for k in wanted]


def combined_optimizer(kernels) -> CombinedOptimizer:
    return CombinedOptimizer([(f"group{i}", k.fused_optimizer) for i, k in enumerate(kernels)])
