"""`dynamicemb.utils` of the plugin surface (reference utils.py:28-60): the TorchRec module types a model is scanned for
and the dtype mapping of the extension module."""
from typing import Dict, Set, Type

import torch

from dynamicemb_extensions import DynamicEmbDataType



def __getattr__(name):
    """`TORCHREC_TYPES` (reference utils.py:28) resolves TorchRec on first use: `import dynamicemb` itself must work on a box
    without TorchRec (the lookup modules and the benchmark do not touch the plugin surface)."""
    if name == "TORCHREC_TYPES":
        from ._torchrec import EmbeddingBagCollection, EmbeddingCollection

        types: Set[Type] = {EmbeddingBagCollection, EmbeddingCollection}
        globals()["TORCHREC_TYPES"] = types
        return types
    raise AttributeError(name)


DTYPE_NUM_BYTES: Dict[torch.dtype, int] = {torch.float32: 4, torch.float16: 2, torch.bfloat16: 2}

_TORCH_TO_DYN = {torch.float32: DynamicEmbDataType.Float32, torch.bfloat16: DynamicEmbDataType.BFloat16,
                 torch.float16: DynamicEmbDataType.Float16, torch.int64: DynamicEmbDataType.Int64,
                 torch.uint64: DynamicEmbDataType.UInt64, torch.int32: DynamicEmbDataType.Int32,
                 torch.uint32: DynamicEmbDataType.UInt32}


def torch_to_dyn_emb(torch_dtype: torch.dtype) -> DynamicEmbDataType:
    if torch_dtype not in _TORCH_TO_DYN:
        raise ValueError(f"Unsupported torch dtype: {torch_dtype}")
    return _TORCH_TO_DYN[torch_dtype]


def tabulate(table, headers=None, sub_headers: bool = False) -> str:
    """plain-text table (first row = headers when none are given)"""
    if headers is None:
        headers, table = table[0], table[1:]
    widths = [max(len(str(h)), *(len(str(r[i])) for r in table)) if table else len(str(h)) for i, h in enumerate(headers)]
    lines = [" | ".join(str(h).center(w) for h, w in zip(headers, widths)), " | ".join("-" * w for w in widths)]
    lines += [" | ".join(str(c).ljust(w) for c, w in zip(r, widths)) for r in table]
    if sub_headers and len(lines) > 3:
        lines.insert(3, " | ".join("-" * w for w in widths))
    return "\n".join(lines)
