"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

ctypes front end of ``liboracle.so`` (integer/byte restatements in C,
``demb_oracle.c``) plus numpy restatements of the floating point pieces of the
DynamicEmb lookup path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path in
``recsys-examples_amd/`` never does.

Parity status: "unpinned" against the reference's CUDA binaries (they cannot be
built or run here and the reference ships no golden files for this path);
pinned against the reference's own Python hash copy, its test invariants and
its DEBUG-initializer closed forms (see tests/test_oracle_demb.py).

Citations are relative to /root/reference/corelib/dynamicemb/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

POLICY_CONST, POLICY_ASSIGN, POLICY_ACCUMULATE, POLICY_GLOBAL_TIMER, POLICY_LRU_LFU = range(5)
RES_INSERT, RES_RECLAIM, RES_ASSIGN, RES_EVICT, RES_DUPLICATED, RES_BUSY, RES_ILLEGAL, RES_INIT = range(8)

EMPTY_KEY = 0xFFFFFFFFFFFFFFFF
RECLAIM_KEY = 0xFFFFFFFFFFFFFFFE
LOCKED_KEY = 0xFFFFFFFFFFFFFFFD


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "demb_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-o", _LIB_PATH, src], cwd=_HERE
        )
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_fmix64.restype = ctypes.c_uint64
        _lib.orc_fmix64.argtypes = [ctypes.c_uint64]
        _lib.orc_hash.restype = ctypes.c_int64
        _lib.orc_hash.argtypes = [ctypes.c_uint64]
        _lib.orc_digest.restype = ctypes.c_uint8
        _lib.orc_digest.argtypes = [ctypes.c_uint64]
        _lib.orc_empty_digest.restype = ctypes.c_uint8
        _lib.orc_bucketize_keys.restype = ctypes.c_int64
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _i64(x):
    return ctypes.c_int64(int(x))


def _u64(x):
    return ctypes.c_uint64(int(x) & 0xFFFFFFFFFFFFFFFF)


def fmix64(k: int) -> int:
    return int(lib().orc_fmix64(_u64(k)))


def hash64(k: int) -> int:
    return int(lib().orc_hash(_u64(k)))


def digest(k: int) -> int:
    return int(lib().orc_digest(_u64(k)))


def empty_digest() -> int:
    return int(lib().orc_empty_digest())


def _keys(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype == np.int64:
        a = a.view(np.uint64)
    assert a.dtype == np.uint64
    return a


class OracleTable:
    """``LinearBucketTable`` (scored_hashtable.py:294-474) restated on numpy."""

    def __init__(self, capacities, bucket_capacity: int = 128, num_scores: int = 1, enable_overflow: bool = False):
        C = ((bucket_capacity + 15) // 16) * 16  # scored_hashtable.py:362-375
        self.C, self.ns = C, num_scores
        nb = [(c + C - 1) // C for c in capacities]  # :381-393
        self.tbo = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        self.num_buckets = int(self.tbo[-1])
        self.capacity = self.num_buckets * C
        self.per_table_capacity = [n * C for n in nb]
        self.storage = np.empty(self.num_buckets * C * (9 + 8 * num_scores), dtype=np.uint8)
        self.bucket_sizes = np.zeros(self.num_buckets, dtype=np.int32)
        self.counter = np.zeros(self.capacity, dtype=np.int32)
        self._lock = np.zeros(max(self.capacity, 1), dtype=np.uint8)
        lib().orc_table_init(_p(self.storage), _i64(self.num_buckets), _i64(C), _i64(num_scores))
        # overflow region (scored_hashtable.py:426-474): one bucket of 3*C slots per logical table, ref-counter tail
        self.enable_overflow = enable_overflow
        if enable_overflow:
            T = len(capacities)
            self.ocap = 3 * C
            self.ovf_storage = np.empty(T * self.ocap * (9 + 8 * num_scores), dtype=np.uint8)
            lib().orc_table_init(_p(self.ovf_storage), _i64(T), _i64(self.ocap), _i64(num_scores))
            self.ovf_sizes = np.zeros(T, dtype=np.int32)
            self.out_off = np.asarray(self.per_table_capacity, dtype=np.int64)
            self.counter = np.zeros(self.capacity + T * self.ocap, dtype=np.int32)
            self._ovf_lock = np.zeros(T * self.ocap, dtype=np.uint8)

    def counter_index(self, slots, table_ids):
        """flat ref-counter index of table-relative slots (update_counter_with_layout_kernel, insert_and_evict.cu:27-58)"""
        slots = np.asarray(slots, np.int64)
        tids = np.asarray(table_ids, np.int64)
        flat = self.tbo[tids] * self.C + slots
        if self.enable_overflow:
            per = self.out_off[tids]
            flat = np.where(slots < per, flat, self.capacity + tids * self.ocap + (slots - per))
        return flat

    def lookup_ovf(self, keys, table_ids, score_in=None, policy=POLICY_CONST, timer=0):
        keys = _keys(keys)
        n = keys.size
        tids = np.ascontiguousarray(table_ids, dtype=np.int64)
        so = np.empty(n, np.int64)
        fo = np.empty(n, np.uint8)
        idx = np.empty(n, np.int64)
        si = None if score_in is None else _keys(score_in)
        lib().orc_table_lookup_ovf(_p(self.storage), _p(self.tbo), _i64(self.C), _i64(self.ns), _p(self.ovf_storage),
                                   _i64(self.ocap), _p(self.out_off), _i64(n), _p(keys), _p(tids), _p(si),
                                   ctypes.c_int(policy), _u64(timer), _p(so), _p(fo), _p(idx))
        return so, fo.astype(bool), idx

    def insert_ovf(self, keys, table_ids, score_in=None, policy=POLICY_ASSIGN, timer=0):
        """insert_and_evict_with_counter_and_overflow -> (indices, results, score_out, (ev_keys, ev_idx, ev_scores, ev_tids))"""
        keys = _keys(keys)
        n = keys.size
        tids = np.ascontiguousarray(table_ids, dtype=np.int64)
        idx = np.empty(n, np.int64)
        res = np.empty(n, np.uint8)
        so = np.empty(n, np.int64)
        si = None if score_in is None else _keys(score_in)
        nev = np.zeros(1, np.int64)
        ek, ei, es, et = np.empty(n, np.uint64), np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int64)
        ovf_counter = self.counter[self.capacity:]
        lib().orc_table_insert_ovf(_p(self.storage), _p(self.tbo), _i64(self.C), _i64(self.ns), _p(self.bucket_sizes),
                                   _p(self.counter), _p(self._lock), _p(self.ovf_storage), _i64(self.ocap),
                                   _p(self.ovf_sizes), _p(ovf_counter), _p(self.out_off), _p(self._ovf_lock), _i64(n),
                                   _p(keys), _p(tids), _p(si), ctypes.c_int(policy), _u64(timer), _p(idx), _p(res),
                                   _p(so), _p(nev), _p(ek), _p(ei), _p(es), _p(et))
        m = int(nev[0])
        return idx, res, so, (ek[:m], ei[:m], es[:m], et[:m])

    # views (table_partition, src/table_operation/table.cu:21-65)
    def _view(self):
        C, ns = self.C, self.ns
        b = self.storage.reshape(self.num_buckets, C * (9 + 8 * ns))
        keys = b[:, : 8 * C].copy().view(np.uint64)
        dig = b[:, 8 * C : 9 * C].copy()
        scores = b[:, 9 * C :].copy().view(np.uint64).reshape(self.num_buckets, C, ns)
        return keys, dig, scores

    def lookup(self, keys, table_ids, score_in=None, policy=POLICY_CONST, timer=0):
        keys = _keys(keys)
        n = keys.size
        tids = np.ascontiguousarray(table_ids, dtype=np.int64)
        so = np.empty(n, np.int64)
        fo = np.empty(n, np.uint8)
        idx = np.empty(n, np.int64)
        si = None if score_in is None else _keys(score_in)
        lib().orc_table_lookup(_p(self.storage), _p(self.tbo), _i64(self.C), _i64(self.ns), _i64(n),
                               _p(keys), _p(tids), _p(si), ctypes.c_int(policy), _u64(timer),
                               _p(so), _p(fo), _p(idx))
        return so, fo.astype(bool), idx

    def insert(self, keys, table_ids, score_in=None, policy=POLICY_ASSIGN, timer=0, evict_out=False):
        """One call == one kernel launch of table_insert(_and_evict) + unlock."""
        keys = _keys(keys)
        n = keys.size
        tids = np.ascontiguousarray(table_ids, dtype=np.int64)
        idx = np.empty(n, np.int64)
        res = np.empty(n, np.uint8)
        so = np.empty(n, np.int64)
        si = None if score_in is None else _keys(score_in)
        nev = np.zeros(1, np.int64)
        ek = np.empty(n, np.uint64) if evict_out else None
        ei = np.empty(n, np.int64) if evict_out else None
        es = np.empty(n, np.int64) if evict_out else None
        et = np.empty(n, np.int64) if evict_out else None
        lib().orc_table_insert(_p(self.storage), _p(self.tbo), _i64(self.C), _i64(self.ns),
                               _p(self.bucket_sizes), _p(self.counter), _p(self._lock), _i64(n),
                               _p(keys), _p(tids), _p(si), ctypes.c_int(policy), _u64(timer),
                               _p(idx), _p(res), _p(so), _p(nev), _p(ek), _p(ei), _p(es), _p(et))
        if evict_out:
            m = int(nev[0])
            return idx, res, so, (ek[:m], ei[:m], es[:m], et[:m])
        return idx, res, so

    def bucketize(self, keys, table_ids):
        keys = _keys(keys)
        n = keys.size
        tids = np.ascontiguousarray(table_ids, dtype=np.int64)
        ko = np.empty(n, np.uint64)
        off = np.zeros(n + 1, np.int64)
        inv = np.empty(n, np.int64)
        nb = lib().orc_bucketize_keys(_p(self.tbo), _i64(self.C), _i64(n), _p(keys), _p(tids),
                                      _p(ko), _p(off), _p(inv))
        return ko, off[: nb + 1], inv

    def insert_deterministic(self, keys, table_ids, score_in=None, policy=POLICY_ASSIGN, timer=0,
                             evict_out=False):
        """DEMB_DETERMINISM_MODE (scored_hashtable.py:1451-1558): sort by (bucket, key), insert
        wave i = the i-th key of every bucket, then one CONST lookup for the final indices."""
        keys = _keys(keys)
        tids = np.ascontiguousarray(table_ids, dtype=np.int64)
        ko, off, inv = self.bucketize(keys, tids)
        lens = np.diff(off)
        evs = [[], [], [], []]
        for w in range(int(lens.max()) if lens.size else 0):
            sel = off[:-1][lens > w] + w
            k = ko[sel]
            t = tids[inv[sel]]
            s = None if score_in is None else _keys(score_in)[inv[sel]]
            out = self.insert(k, t, s, policy, timer, evict_out=evict_out)
            if evict_out:
                ek, ei, es, et = out[3]
                # Busy entries carry -(i+1) relative to the wave; keep as is (reference does too)
                for lst, a in zip(evs, (ek, ei, es, et)):
                    lst.append(a)
        _, _, idx = self.lookup(keys, tids, None, POLICY_CONST)
        if evict_out:
            cat = [np.concatenate(x) if x else np.empty(0, np.int64) for x in evs]
            return idx, tuple(cat)
        return idx

    def erase(self, keys, table_ids):
        keys = _keys(keys)
        tids = np.ascontiguousarray(table_ids, dtype=np.int64)
        idx = np.empty(keys.size, np.int64)
        lib().orc_table_erase(_p(self.storage), _p(self.tbo), _i64(self.C), _i64(self.ns),
                              _p(self.bucket_sizes), _i64(keys.size), _p(keys), _p(tids), _p(idx))
        return idx


def segmented_unique(keys, seg_range, in_freq=None, count_freq=False):
    keys = _keys(keys)
    n = keys.size
    seg = np.ascontiguousarray(seg_range, dtype=np.int64)
    T = seg.size - 1
    uk = np.empty(n, np.uint64)
    oi = np.empty(n, np.int64)
    to = np.empty(T + 1, np.int64)
    fr = np.zeros(n, np.int64) if count_freq else None
    inf = None if in_freq is None else np.ascontiguousarray(in_freq, dtype=np.int64)
    lib().orc_segmented_unique(_i64(n), _p(keys), _p(seg), _i64(T), _p(inf), ctypes.c_int(int(count_freq)),
                               _p(uk), _p(oi), _p(to), _p(fr))
    nu = int(to[-1])
    return uk[:nu], oi, to, (fr[:nu] if count_freq else None)


def expand_table_ids(offsets, n):
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    out = np.empty(n, np.int64)
    lib().orc_expand_table_ids(_p(off), _i64(off.size - 1), _i64(n), _p(out))
    return out


def get_table_range(offsets, feature_offsets, B=None):
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    fo = np.ascontiguousarray(feature_offsets, dtype=np.int64)
    T = fo.size - 1
    if B is None:
        B = (off.size - 1) // int(fo[-1])
    out = np.empty(T + 1, np.int64)
    lib().orc_get_table_range(_p(off), _p(fo), _i64(T), _i64(B), _p(out))
    return out


def block_bucketize(offsets, indices, W, B, block_sizes, dist_type):
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    idx = _keys(indices)
    FB = off.size - 1
    bs = np.ascontiguousarray(block_sizes, dtype=np.int64)
    nl = np.empty(W * FB, np.int64)
    no = np.empty(W * FB + 1, np.int64)
    ni = np.empty(idx.size, np.uint64)
    perm = np.empty(idx.size, np.int64)
    lib().orc_block_bucketize(_i64(W), _i64(FB), _i64(B), _p(off), _p(idx), _p(bs), ctypes.c_int(dist_type),
                              _p(nl), _p(no), _p(ni), _p(perm))
    return nl, no, ni, perm


def block_bucketize_ex(offsets, indices, W, B, block_sizes, dist_types=None, bag_feature=None, pos=None):
    """orc_block_bucketize_ex: per-feature dist types, variable batch size per feature (bag_feature [FB]) and uneven shard
    boundaries (pos: one sorted int64 array of W + 1 boundaries per feature)."""
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    idx = _keys(indices)
    FB = off.size - 1
    bs = np.ascontiguousarray(block_sizes, dtype=np.int64)
    dt_ = None if dist_types is None else np.ascontiguousarray(dist_types, dtype=np.int32)
    bf = None if bag_feature is None else np.ascontiguousarray(bag_feature, dtype=np.int64)
    pc = po = None
    if pos is not None:
        pc = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int64) for x in pos]), dtype=np.int64)
        po = np.concatenate([[0], np.cumsum([len(x) for x in pos])]).astype(np.int64)
    nl = np.empty(W * FB, np.int64)
    no = np.empty(W * FB + 1, np.int64)
    ni = np.empty(idx.size, np.uint64)
    perm = np.empty(idx.size, np.int64)
    none = ctypes.c_void_p(0)
    lib().orc_block_bucketize_ex(_i64(W), _i64(FB), _i64(B), _p(off), _p(idx), _p(bs), _p(dt_) if dt_ is not None else none,
                                 _p(bf) if bf is not None else none, _p(pc) if pc is not None else none,
                                 _p(po) if po is not None else none, _p(nl), _p(no), _p(ni), _p(perm))
    return nl, no, ni, perm


def debug_init(keys, dim):
    keys = _keys(keys)
    out = np.empty((keys.size, dim), np.float32)
    lib().orc_debug_init(_i64(keys.size), _i64(dim), _i64(dim), _p(keys), _p(out))
    return out


# --------------------------------------------------------------------------
# floating point restatements (numpy, fp32 arithmetic like the kernels)
# --------------------------------------------------------------------------

def round_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even) -> fp32, like a `(bf16)` cast."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    nan = np.isnan(x)
    out = r.astype(np.uint32).view(np.float32).copy()
    out[nan] = np.nan
    return out


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    if dtype in ("f32", "float32"):
        return np.ascontiguousarray(x, dtype=np.float32)
    if dtype in ("bf16", "bfloat16"):
        return round_bf16(x)
    if dtype in ("f16", "float16"):
        return x.astype(np.float16).astype(np.float32)
    raise ValueError(dtype)


def gather_rows(table: np.ndarray, slots: np.ndarray, dim: int) -> np.ndarray:
    """load_from_flat(EMBEDDING) (src/dynamic_emb_op.cu:294-347): rows of slot<0 are skipped; this
    build defines them as zeros (the reference leaves them for the initializer / eval zero fill,
    key_value_table.py:2915-2949)."""
    out = np.zeros((slots.size, dim), np.float32)
    ok = slots >= 0
    out[ok] = table[slots[ok], :dim]
    return out


def gather_pooled(unique_embs, reverse_idx, offsets, B, combiner, D_offsets=None, out_dtype="f32"):
    """gather_embedding_pooled (src/dynamic_emb_op.cu:106-133, lookup_kernel.cuh:900-962,
    lookup_forward.cu:30-75): slot i = f*B + b, fp32 sequential sum over the bag, MEAN divides
    by the bag length when it is > 0, output [B, total_D]."""
    FB = offsets.size - 1
    F = FB // B
    D = unique_embs.shape[1]
    if D_offsets is None:
        D_offsets = np.arange(F + 1, dtype=np.int64) * D
    total_D = int(D_offsets[-1])
    out = np.zeros((B, total_D), np.float32)
    for i in range(FB):
        f, b = divmod(i, B)
        d0, d1 = int(D_offsets[f]), int(D_offsets[f + 1])
        acc = np.zeros(d1 - d0, np.float32)
        for j in range(int(offsets[i]), int(offsets[i + 1])):
            acc = acc + unique_embs[reverse_idx[j], : d1 - d0].astype(np.float32)
        L = int(offsets[i + 1] - offsets[i])
        if combiner == 1 and L > 0:
            acc = acc / np.float32(L)
        out[b, d0:d1] = acc
    return round_to(out, out_dtype)


def gather_pooled_fast(unique_embs, reverse_idx, offsets, B, combiner, out_dtype="f32"):
    """Vectorised variant for uniform D (same arithmetic order: np.add.reduceat adds rows of a bag
    in order)."""
    FB = offsets.size - 1
    F = FB // B
    D = unique_embs.shape[1]
    lens = np.diff(offsets)
    rows = unique_embs[reverse_idx].astype(np.float32)
    acc = np.zeros((FB, D), np.float32)
    nz = lens > 0
    if rows.shape[0]:
        starts = offsets[:-1][nz]
        acc[nz] = np.add.reduceat(rows, starts, axis=0)
    if combiner == 1:
        acc[nz] = acc[nz] / lens[nz].astype(np.float32)[:, None]
    out = acc.reshape(F, B, D).transpose(1, 0, 2).reshape(B, F * D)
    return round_to(out, out_dtype)


def gather_sequence(unique_embs, reverse_idx, out_dtype="f32"):
    """gather_embedding (src/dynamic_emb_op.cu:79-104): out[i] = unique_embs[rev[i]]."""
    return round_to(unique_embs[reverse_idx].astype(np.float32), out_dtype)


def reduce_grads(reverse_idx, grads, num_unique, B=0, offsets=None, D_offsets=None, combiner=-1,
                 out_dtype="f32"):
    """reduce_grads (src/dynamic_emb_op.cu:159-285, lookup_backward.cu:32-374):
    unique_grads[u] = sum_j [rev[j]==u] g(j), fp32 accumulate, result in the grad dtype.
    pooled: g(j) = grads[b_j, Doff[f_j]:...] * (1/len(bag) for MEAN); sequence: grads[j]."""
    n = reverse_idx.size
    if offsets is None:
        D = grads.shape[1]
        out = np.zeros((num_unique, D), np.float32)
        np.add.at(out, reverse_idx, grads.astype(np.float32))
        return round_to(out, out_dtype)
    FB = offsets.size - 1
    F = FB // B
    total_D = grads.shape[1]
    if D_offsets is None:
        D = total_D // F
        D_offsets = np.arange(F + 1, dtype=np.int64) * D
    maxD = int(np.max(np.diff(D_offsets)))
    out = np.zeros((num_unique, maxD), np.float32)
    for i in range(FB):
        f, b = divmod(i, B)
        d0, d1 = int(D_offsets[f]), int(D_offsets[f + 1])
        L = int(offsets[i + 1] - offsets[i])
        scale = np.float32(1.0)
        if combiner == 1 and L > 0:
            scale = np.float32(1.0) / np.float32(L)
        g = grads[b, d0:d1].astype(np.float32) * scale
        for j in range(int(offsets[i]), int(offsets[i + 1])):
            out[reverse_idx[j], : d1 - d0] += g
    return round_to(out, out_dtype)


def reduce_grads_pooled_fast(reverse_idx, grads, num_unique, B, offsets, combiner, out_dtype="f32"):
    FB = offsets.size - 1
    F = FB // B
    D = grads.shape[1] // F
    lens = np.diff(offsets)
    g = grads.astype(np.float32).reshape(B, F, D).transpose(1, 0, 2).reshape(FB, D)
    if combiner == 1:
        sc = np.where(lens > 0, np.float32(1.0) / np.maximum(lens, 1).astype(np.float32), np.float32(1.0))
        g = g * sc[:, None].astype(np.float32)
    per_key = np.repeat(g, lens, axis=0)
    out = np.zeros((num_unique, D), np.float32)
    np.add.at(out, reverse_idx, per_key)
    return round_to(out, out_dtype)


# optimizers: src/optimizer_kernel.cuh:40-411 (fp32 maths); rows = [emb | state...]
def sgd_update(rows, grads, D, lr):
    rows[:, :D] = rows[:, :D] - grads.astype(np.float32) * np.float32(lr)
    return rows


def adam_update(rows, grads, D, lr, beta1, beta2, eps, weight_decay, iter_num):
    g = grads.astype(np.float32)
    b1, b2 = np.float32(beta1), np.float32(beta2)
    w, m, v = rows[:, :D], rows[:, D:2 * D], rows[:, 2 * D:3 * D]
    m[:] = b1 * m + (np.float32(1) - b1) * g
    v[:] = b2 * v + (np.float32(1) - b2) * g * g
    mh = m / (np.float32(1) - np.float32(np.power(np.float64(beta1), iter_num)))
    vh = v / (np.float32(1) - np.float32(np.power(np.float64(beta2), iter_num)))
    w[:] = w - np.float32(lr) * (mh / (np.sqrt(vh) + np.float32(eps)) + np.float32(weight_decay) * w)
    return rows


def adagrad_update(rows, grads, D, lr, eps):
    g = grads.astype(np.float32)
    w, G = rows[:, :D], rows[:, D:2 * D]
    G[:] = G + g * g
    w[:] = w - np.float32(lr) * g / (np.sqrt(G) + np.float32(eps))
    return rows


def rowwise_adagrad_update(rows, grads, D, lr, eps):
    g = grads.astype(np.float32)
    w = rows[:, :D]
    G = rows[:, D] + (g * g).sum(axis=1, dtype=np.float32) / np.float32(D)
    rows[:, D] = G
    w[:] = w - np.float32(lr) * g / (np.sqrt(G)[:, None] + np.float32(eps))
    return rows
