"""DynamicEmbDump / DynamicEmbLoad (reference dump_load.py:103-330): save / restore every dynamic embedding table of a
model in the reference's wire format -- one folder per embedding collection (named by the collection's path in the model),
per table and rank the raw little-endian files `<table>_emb_{keys,values,scores,opt_values}.rank_R.world_size_W` plus
`<table>_opt_args.json` (written / read by BatchedDynamicEmbeddingTablesV2.dump / .load)."""
import os
from typing import Dict, List, Optional, Tuple

import torch.distributed as dist
from torch import nn


def _unwrap(m: nn.Module) -> nn.Module:
    """wrappers that only forward (DistributedModelParallel, DDP, Float16Module) do not show up in a collection's path"""
    while True:
        inner = getattr(m, "_dmp_wrapped_module", None)
        if inner is None and type(m).__name__ in ("DistributedDataParallel", "Float16Module", "DDP"):
            inner = getattr(m, "module", None)
        if not isinstance(inner, nn.Module):
            return m
        m = inner


def find_sharded_modules(model: nn.Module, path: str = "model") -> List[Tuple[str, nn.Module]]:
    """(path, module) of every sharded collection that holds dynamic embedding tables; paths start at "model" and skip
    wrapper modules, as the folder names of the reference's dumps do (dump_load.py:31-50)"""
    model = _unwrap(model)
    if hasattr(model, "dynamic_embedding_modules"):
        return [(path, model)]
    found = []
    for name, child in model.named_children():
        found.extend(find_sharded_modules(child, f"{path}.{name}" if path else name))
    return found


def get_dynamic_emb_module(model: nn.Module) -> List[nn.Module]:
    """every BatchedDynamicEmbeddingTablesV2 under `model`"""
    return [m for _, coll in find_sharded_modules(model) for m in coll.dynamic_embedding_modules()]


def _barrier(pg) -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=pg)


def DynamicEmbDump(path: str, model: nn.Module, table_names: Optional[Dict[str, List[str]]] = None,
                   optim: Optional[bool] = False, counter: Optional[bool] = False, pg=None,
                   allow_overwrite: bool = False) -> None:
    """table_names: {collection path: [table names]} to restrict the dump (default: everything)"""
    colls = find_sharded_modules(model)
    rank = dist.get_rank(group=pg) if dist.is_available() and dist.is_initialized() else 0
    if rank == 0:
        os.makedirs(path, exist_ok=True)
    _barrier(pg)
    for cpath, coll in colls:
        if table_names is not None and cpath not in table_names:
            continue
        folder = os.path.join(path, cpath)
        if rank == 0:
            if os.path.isdir(folder) and os.listdir(folder) and not allow_overwrite:
                raise FileExistsError(f"{folder} is not empty (pass allow_overwrite=True to replace the dump)")
            os.makedirs(folder, exist_ok=True)
        _barrier(pg)
        only = None if table_names is None else table_names[cpath]
        for m in coll.dynamic_embedding_modules():
            m.dump(folder, optim=bool(optim), counter=bool(counter), table_names=only, pg=pg)
    _barrier(pg)


def DynamicEmbLoad(path: str, model: nn.Module, table_names: Optional[Dict[str, List[str]]] = None,
                   optim: bool = False, counter: bool = False, pg=None) -> None:
    """every rank reads every file of a table and keeps the keys it owns (`key % world_size == rank`), so a dump taken
    with one world size loads into another"""
    for cpath, coll in find_sharded_modules(model):
        if table_names is not None and cpath not in table_names:
            continue
        folder = os.path.join(path, cpath)
        if not os.path.isdir(folder):
            raise FileNotFoundError(f"no dump of collection {cpath!r} under {path}")
        only = None if table_names is None else table_names[cpath]
        for m in coll.dynamic_embedding_modules():
            m.load(folder, optim=optim, counter=counter, table_names=only, pg=pg)
    _barrier(pg)
