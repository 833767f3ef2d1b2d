"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

`DictEmbeddingTwin`: what a dynamic embedding module must compute, with none of its machinery: a Python dict per table
(key -> row [embedding | optimizer state], fp32), first-touch initialisation by a closed form, pooled / sequence forward in
float64, fp32 optimizer maths of oracle.py (optimizer_kernel.cuh:40-411 restated).  No hash table, no buckets, no tiers,
no dedup, no prefetch, no sharding -- so storage tiers (HBM / host / hybrid / cache), the prefetch pipeline, the
pre-communication dedup and a dump -> load cycle can all be checked against it instead of against another mode of the
product.  Semantics restated from the reference's module contract (batched_dynamicemb_function.py:559-932,1042-1300,
SURVEY appendix A.3-A.4): training inserts unseen keys with the initialiser (state = initial accumulator value), eval gives
zeros for them; every unique key of a batch is updated exactly once with the sum (MEAN: length-scaled) of its gradients;
Adam's step count is the number of backward calls."""
from typing import Callable, Dict, List, Optional

import numpy as np

from . import oracle as orc


def debug_row(key: int, dim: int) -> np.ndarray:
    """DEBUG initialiser: every element = float(key % 100000) (initializer.cuh:158-176)"""
    return np.full(dim, np.float32(int(key) % 100000), np.float32)


class DictEmbeddingTwin:
    def __init__(self, dims: List[int], feature_table_map: List[int], pooling: str = "SUM", optimizer: str = "sgd",
                 lr: float = 0.01, eps: float = 1e-8, beta1: float = 0.9, beta2: float = 0.999, weight_decay: float = 0.0,
                 init: Callable[[int, int], np.ndarray] = debug_row, state_init: float = 0.0):
        self.dims, self.fmap, self.pooling, self.opt = list(dims), list(feature_table_map), pooling, optimizer
        self.hp = dict(lr=lr, eps=eps, beta1=beta1, beta2=beta2, weight_decay=weight_decay)
        self.init, self.state_init = init, state_init
        self.tables: List[Dict[int, np.ndarray]] = [dict() for _ in dims]
        self.iter = 0
        self._last = None

    def _state_dim(self, d: int) -> int:
        return {"sgd": 0, "adam": 2 * d, "adagrad": d, "rowwise_adagrad": 4}[self.opt]

    def _row(self, t: int, key: int, create: bool) -> Optional[np.ndarray]:
        r = self.tables[t].get(int(key))
        if r is None and create:
            d = self.dims[t]
            r = np.concatenate([self.init(int(key), d), np.full(self._state_dim(d), np.float32(self.state_init), np.float32)])
            self.tables[t][int(key)] = r
        return r

    def forward(self, keys: np.ndarray, offsets: np.ndarray, train: bool = True) -> np.ndarray:
        F = len(self.fmap)
        B = (len(offsets) - 1) // F
        self._last = (np.asarray(keys).copy(), np.asarray(offsets).copy(), B)
        if self.pooling == "NONE":
            d = self.dims[0]
            out = np.zeros((len(keys), d), np.float64)
            for i in range(F * B):
                t = self.fmap[i // B]
                for j in range(offsets[i], offsets[i + 1]):
                    r = self._row(t, keys[j], train)
                    if r is not None:
                        out[j] = r[:d]
            return out
        col = np.concatenate([[0], np.cumsum([self.dims[t] for t in self.fmap])])
        out = np.zeros((B, col[-1]), np.float64)
        for i in range(F * B):
            f, b = divmod(i, B)
            t, d = self.fmap[f], self.dims[self.fmap[f]]
            n = offsets[i + 1] - offsets[i]
            acc = np.zeros(d, np.float64)
            for j in range(offsets[i], offsets[i + 1]):
                r = self._row(t, keys[j], train)
                if r is not None:
                    acc += r[:d]
            out[b, col[f]:col[f + 1]] = acc / n if (self.pooling == "MEAN" and n > 0) else acc
        return out

    def backward(self, grads: np.ndarray) -> None:
        keys, offsets, B = self._last
        F = len(self.fmap)
        self.iter += 1
        col = np.concatenate([[0], np.cumsum([self.dims[t] for t in self.fmap])])
        sums: List[Dict[int, np.ndarray]] = [dict() for _ in self.dims]
        for i in range(F * B):
            f, b = divmod(i, B)
            t, d = self.fmap[f], self.dims[self.fmap[f]]
            n = offsets[i + 1] - offsets[i]
            for j in range(offsets[i], offsets[i + 1]):
                g = grads[j, :d] if self.pooling == "NONE" else grads[b, col[f]:col[f + 1]]
                g = np.asarray(g, np.float64) / (n if self.pooling == "MEAN" else 1)
                k = int(keys[j])
                sums[t][k] = sums[t].get(k, 0.0) + g
        for t, per_key in enumerate(sums):
            d = self.dims[t]
            for k, g in per_key.items():
                r = self.tables[t].get(k)
                if r is None:
                    continue
                row, gg = r[None, :].copy(), np.asarray(g, np.float32)[None, :]
                if self.opt == "sgd":
                    orc.sgd_update(row, gg, d, self.hp["lr"])
                elif self.opt == "adam":
                    orc.adam_update(row, gg, d, self.hp["lr"], self.hp["beta1"], self.hp["beta2"], self.hp["eps"],
                                    self.hp["weight_decay"], self.iter)
                elif self.opt == "adagrad":
                    orc.adagrad_update(row, gg, d, self.hp["lr"], self.hp["eps"])
                else:
                    orc.rowwise_adagrad_update(row, gg, d, self.hp["lr"], self.hp["eps"])
                self.tables[t][k] = row[0]

    def rows(self, t: int, keys) -> (np.ndarray, np.ndarray):
        d = self.dims[t]
        found = np.array([int(k) in self.tables[t] for k in keys], bool)
        out = np.zeros((len(keys), d), np.float32)
        for i, k in enumerate(keys):
            if found[i]:
                out[i] = self.tables[t][int(k)][:d]
        return found, out
