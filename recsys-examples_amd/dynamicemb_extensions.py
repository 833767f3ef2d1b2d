"""Drop-in for the reference's pybind11 module ``dynamicemb_extensions``
(corelib/dynamicemb/src/module_bind.cu:33-44): same function names, argument order and meaning,
same per-key error behaviour (index -1 / InsertResult as data, exceptions only for bad arguments),
but every op is a hand-written gfx950 kernel behind the C ABI of include/recsys_amd.h.

Tensors are only carriers of device pointers here; there is no PyTorch compute and no fallback.
Reference citations are relative to /root/reference/corelib/dynamicemb/.
"""
from __future__ import annotations

import ctypes
import enum
from typing import List, Optional, Sequence, Tuple

import torch

import mi355_native as N
from mi355_native import c_f, c_i64, c_int, c_p, c_u64, check, dt, lib, ptr, stream


# ---------------------------------------------------------------- enums (table.cu:186-204) ----
class ScorePolicy(enum.IntEnum):
    CONST = 0
    ASSIGN = 1
    ACCUMULATE = 2
    GLOBAL_TIMER = 3
    LRU_LFU = 4


class InsertResult(enum.IntEnum):
    INSERT = 0
    RECLAIM = 1
    ASSIGN = 2
    EVICT = 3
    DUPLICATED = 4
    BUSY = 5
    ILLEGAL = 6
    INIT = 7


class DynamicEmbDataType(enum.IntEnum):  # dynamic_emb_op.cu:803-812
    Float32 = 0
    BFloat16 = 1
    Float16 = 2
    Int64 = 3
    UInt64 = 4
    Int32 = 5
    UInt32 = 6
    Size_t = 7


class EvictStrategy(enum.IntEnum):  # dynamic_emb_op.cu:814-820
    KLru = 0
    KLfu = 1
    KEpochLru = 2
    KEpochLfu = 3
    KCustomized = 4


class OptimizerType(enum.IntEnum):
    SGD = 1
    ADAM = 2
    ADAGRAD = 3
    ROWWISE_ADAGRAD = 4


_BYTES = {torch.int64: 8, torch.uint64: 8, torch.uint8: 1, torch.int32: 4, torch.float32: 4,
          torch.bfloat16: 2, torch.float16: 2, torch.bool: 1}

# test hook: when non-zero, GLOBAL_TIMER / LRU_LFU use this value instead of the device clock
TIMER_OVERRIDE = 0


def _u8(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.view(torch.uint8) if t.dtype == torch.bool else t


def _i64dev(x, device):
    return x


# --------------------------------------------------------------- table views (table.cu:21-87) ----
def table_partition(storage: torch.Tensor, dtypes: Sequence[torch.dtype], bucket_capacity: int, num_buckets: int):
    """Strided [num_buckets, bucket_capacity] views of the SoA bucket arena (table.cu:21-65)."""
    sizes = [_BYTES[d] for d in dtypes]
    bucket_bytes = sum(sizes) * bucket_capacity
    if bucket_bytes * num_buckets != storage.numel() * storage.element_size():
        raise RuntimeError("Storage size mismatched with bucket_bytes * num_buckets")
    out = []
    off = 0
    flat = storage.view(torch.uint8)
    for d, sz in zip(dtypes, sizes):
        if num_buckets == 0:
            out.append(torch.empty((0, bucket_capacity), dtype=d, device=storage.device))
        else:
            v = flat[off:].view(d) if (flat.numel() - off) % sz == 0 else flat[off: off + ((flat.numel() - off) // sz) * sz].view(d)
            out.append(torch.as_strided(v, (num_buckets, bucket_capacity), (bucket_bytes // sz, 1)))
        off += sz * bucket_capacity
    return out


def tensor_partition(input: torch.Tensor, byte_range: Sequence[int], dtypes: Sequence[torch.dtype]):
    flat = input.view(torch.uint8)
    return [flat[byte_range[i]: byte_range[i + 1]].view(d) for i, d in enumerate(dtypes)]


# ------------------------------------------------------------------------------ table ops ----
def table_init(table_storage, bucket_capacity: int, num_buckets: int, num_scores: int = 1):
    check(lib().mi355_table_init(ptr(table_storage), num_buckets, bucket_capacity, num_scores, stream()), "table_init")


def table_lookup(table_storage, table_bucket_offsets, bucket_capacity, keys, table_ids, score_input, policy_type,
                 ovf_storage=None, ovf_bucket_capacity=0, ovf_output_offsets=None, num_scores=1, n_dev=None):
    """table_lookup (lookup.cu:151-191) -> (score_out i64[N], founds bool[N], indices i64[N])."""
    n = keys.numel()
    dev = keys.device
    score_out = torch.empty(n, dtype=torch.int64, device=dev)
    founds = torch.empty(n, dtype=torch.bool, device=dev)
    indices = torch.empty(n, dtype=torch.int64, device=dev)
    if ovf_storage is not None:   # lookup.cu:82-150: main table, then the table's overflow bucket
        check(lib().mi355_table_lookup_overflow(ptr(table_storage), ptr(table_bucket_offsets), bucket_capacity, num_scores, n,
                                                ptr(n_dev), ptr(keys), ptr(table_ids), ptr(score_input), int(policy_type),
                                                c_u64(TIMER_OVERRIDE), ptr(ovf_storage), ovf_bucket_capacity,
                                                ptr(ovf_output_offsets), ptr(score_out), ptr(_u8(founds)), ptr(indices),
                                                stream()), "table_lookup (overflow)")
        return score_out, founds, indices
    check(lib().mi355_table_lookup(ptr(table_storage), ptr(table_bucket_offsets), bucket_capacity, num_scores, n,
                                   ptr(n_dev), ptr(keys), ptr(table_ids), ptr(score_input), int(policy_type),
                                   c_u64(TIMER_OVERRIDE), ptr(score_out), ptr(_u8(founds)), ptr(indices), stream()),
          "table_lookup")
    return score_out, founds, indices


def _insert(table_storage, table_bucket_offsets, bucket_capacity, bucket_sizes, keys, table_ids, score_input,
            policy_type, counter, insert_results, score_output, num_scores, evict, skip=None, indices=None, n_dev=None):
    n = keys.numel()
    dev = keys.device
    if indices is None:
        indices = torch.empty(n, dtype=torch.int64, device=dev)
    ev = [None] * 5
    if evict:
        ev = [torch.zeros(1, dtype=torch.int64, device=dev), torch.empty_like(keys),
              torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev),
              torch.empty(n, dtype=torch.int64, device=dev)]
    check(lib().mi355_table_insert(ptr(table_storage), ptr(table_bucket_offsets), bucket_capacity, num_scores,
                                   ptr(bucket_sizes), ptr(counter), n, ptr(n_dev), ptr(keys), ptr(table_ids),
                                   ptr(score_input), int(policy_type), c_u64(TIMER_OVERRIDE), ptr(_u8(skip)),
                                   ptr(indices), ptr(insert_results), ptr(score_output), ptr(ev[0]), ptr(ev[1]),
                                   ptr(ev[2]), ptr(ev[3]), ptr(ev[4]), stream()), "table_insert")
    return indices, ev


def table_insert(table_storage, table_bucket_offsets, bucket_capacity, bucket_sizes, keys, table_ids, score_input,
                 policy_type, counter, insert_results=None, score_output=None, num_scores=1, skip=None, indices=None,
                 n_dev=None):
    """table_insert (insert.cu) -> indices.  `skip`/`indices`/`n_dev` are extensions of this build."""
    idx, _ = _insert(table_storage, table_bucket_offsets, bucket_capacity, bucket_sizes, keys, table_ids, score_input,
                     policy_type, counter, insert_results, score_output, num_scores, False, skip, indices, n_dev)
    return idx


def table_insert_and_evict(table_storage, table_bucket_offsets, bucket_capacity, bucket_sizes, keys, table_ids,
                           score_input, policy_type, counter, insert_results=None, score_output=None,
                           ovf_storage=None, ovf_bucket_capacity=0, ovf_bucket_sizes=None, ovf_counter=None,
                           ovf_output_offsets=None, num_scores=1):
    """table_insert_and_evict (insert_and_evict.cu) -> (indices, num_evicted[1] (device), evicted_keys,
    evicted_indices, evicted_scores (uint64 bits as in the reference), evicted_table_ids)."""
    if ovf_storage is not None:   # insert_and_evict.cu:201-395
        n = keys.numel()
        dev = keys.device
        indices = torch.empty(n, dtype=torch.int64, device=dev)
        if insert_results is None:
            insert_results = torch.empty(n, dtype=torch.uint8, device=dev)
        ev = [torch.zeros(1, dtype=torch.int64, device=dev), torch.empty_like(keys),
              torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev),
              torch.empty(n, dtype=torch.int64, device=dev)]
        ws = torch.empty(int(lib().mi355_table_insert_overflow_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        check(lib().mi355_table_insert_overflow(
            ptr(table_storage), ptr(table_bucket_offsets), bucket_capacity, num_scores, ptr(bucket_sizes), ptr(counter), n,
            None, ptr(keys), ptr(table_ids), ptr(score_input), int(policy_type), c_u64(TIMER_OVERRIDE), None,
            ptr(ovf_storage), ovf_bucket_capacity, ptr(ovf_bucket_sizes), ptr(ovf_counter), ptr(ovf_output_offsets),
            ptr(indices), ptr(insert_results), ptr(score_output), ptr(ev[0]), ptr(ev[1]), ptr(ev[2]), ptr(ev[3]), ptr(ev[4]),
            ptr(ws), ws.numel(), stream()), "table_insert_and_evict (overflow)")
        return indices, ev[0], ev[1], ev[2], ev[3], ev[4]
    idx, ev = _insert(table_storage, table_bucket_offsets, bucket_capacity, bucket_sizes, keys, table_ids, score_input,
                      policy_type, counter, insert_results, score_output, num_scores, True)
    return idx, ev[0], ev[1], ev[2], ev[3], ev[4]


def table_erase(table_storage, table_bucket_offsets, bucket_capacity, bucket_sizes, keys, table_ids, indices=None,
                num_scores=1):
    check(lib().mi355_table_erase(ptr(table_storage), ptr(table_bucket_offsets), bucket_capacity, num_scores,
                                  ptr(bucket_sizes), keys.numel(), ptr(keys), ptr(table_ids), ptr(indices), stream()),
          "table_erase")


def table_update_counter_with_layout(counter, slot_indices, delta, table_bucket_offsets, bucket_capacity,
                                     main_capacity, num_tables, table_ids=None, overflow_output_offsets=None,
                                     overflow_bucket_capacity=0, n_dev=None):
    if overflow_output_offsets is not None:
        tids = table_ids if table_ids is not None else torch.zeros_like(slot_indices)
        check(lib().mi355_table_update_counter_overflow(ptr(counter), counter.numel(), ptr(slot_indices),
                                                        slot_indices.numel(), ptr(n_dev), int(delta), ptr(tids),
                                                        ptr(table_bucket_offsets), bucket_capacity, main_capacity,
                                                        ptr(overflow_output_offsets), overflow_bucket_capacity, stream()),
              "table_update_counter (overflow)")
        return
    use_layout = table_ids is not None and num_tables > 1
    check(lib().mi355_table_update_counter(ptr(counter), counter.numel(), ptr(slot_indices), slot_indices.numel(),
                                           ptr(n_dev), int(delta), ptr(table_ids if use_layout else None),
                                           ptr(table_bucket_offsets if use_layout else None), bucket_capacity,
                                           stream()), "table_update_counter")


def _gpu(t: torch.Tensor):
    """device for the outputs of an op on `t`: a pinned host table is driven from the current GPU"""
    return t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())


def _num_buckets(table_storage, bucket_capacity, num_scores):
    return table_storage.numel() * table_storage.element_size() // (bucket_capacity * (9 + 8 * num_scores))


def table_export_batch(table_storage, bucket_capacity, batch, offset, key_dtype=torch.int64, threshold=None,
                       table_begin=0, num_scores=1, score_index=0):
    """table_export_batch (export_batch.cu:88-124) -> (counter i64[1], keys[batch], scores i64[batch], indices i64[batch]);
    the first `counter` entries are valid.  Slot order (deterministic); the reference's order is atomic order."""
    dev = _gpu(table_storage)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    keys = torch.empty(batch, dtype=key_dtype, device=dev)
    score = torch.empty(batch, dtype=torch.int64, device=dev)
    indices = torch.empty(batch, dtype=torch.int64, device=dev)
    if batch == 0:
        return counter, keys, score, indices
    if keys.element_size() != 8:
        raise RuntimeError("only 64-bit keys are supported")
    nb = _num_buckets(table_storage, bucket_capacity, num_scores)
    if offset + batch > nb * bucket_capacity:
        raise ValueError("Offset and batch size overflow.")
    ws = _workspace(lib().mi355_table_export_batch_workspace_bytes(batch), dev)
    thr = 0 if threshold is None else int(threshold) & 0xFFFFFFFFFFFFFFFF
    check(lib().mi355_table_export_batch(ptr(table_storage), nb, bucket_capacity, num_scores, batch, offset,
                                         int(threshold is not None), c_u64(thr), table_begin, score_index, ptr(counter),
                                         ptr(keys), ptr(score), ptr(indices), ptr(ws), ws.numel(), stream()),
          "table_export_batch")
    return counter, keys, score, indices


def table_count_matched(table_storage, key_dtype, bucket_capacity, threshold, begin=-1, end=-1, num_scores=1,
                        score_index=0):
    """table_count_matched (count_matched.cu:77-93) -> i64[1] on the device."""
    out = torch.zeros(1, dtype=torch.int64, device=_gpu(table_storage))
    nb = _num_buckets(table_storage, bucket_capacity, num_scores)
    check(lib().mi355_table_count_matched(ptr(table_storage), nb, bucket_capacity, num_scores,
                                          c_u64(int(threshold) & 0xFFFFFFFFFFFFFFFF), begin, end, score_index, ptr(out),
                                          stream()), "table_count_matched")
    return out


def table_copy_score_blocks(src_storage, src_bucket_capacity, dst_storage, dst_bucket_capacity, num_scores,
                            src_bkt_begin, dst_bkt_begin, src_slots, dst_slots, key_dtype=torch.int64):
    n = src_slots.size(0)
    if n == 0:
        return
    check(lib().mi355_table_score_blocks(0, ptr(src_storage), src_bucket_capacity, src_bkt_begin, ptr(dst_storage),
                                         dst_bucket_capacity, dst_bkt_begin, num_scores, n, ptr(src_slots.contiguous()),
                                         ptr(dst_slots.contiguous()), None, stream()), "table_copy_score_blocks")


def table_gather_score_blocks(table_storage, bucket_capacity, num_scores, bkt_begin, slots, key_dtype=torch.int64):
    n = slots.size(0)
    out = torch.empty(n, num_scores, dtype=torch.int64, device=_gpu(table_storage))
    if n:
        check(lib().mi355_table_score_blocks(1, ptr(table_storage), bucket_capacity, bkt_begin, None, 0, 0, num_scores, n,
                                             ptr(slots.contiguous()), None, ptr(out), stream()),
              "table_gather_score_blocks")
    return out


def table_scatter_score_blocks(table_storage, bucket_capacity, num_scores, bkt_begin, slots, values,
                               key_dtype=torch.int64):
    n = slots.size(0)
    if n == 0:
        return
    vals = values.contiguous()
    check(lib().mi355_table_score_blocks(2, None, 0, 0, ptr(table_storage), bucket_capacity, bkt_begin, num_scores, n, None,
                                         ptr(slots.contiguous()), ptr(vals), stream()), "table_scatter_score_blocks")


class _CudaArrayView:
    """zero-copy holder of a raw device-addressable range for torch.as_tensor (CUDA array interface v2, bytes)"""

    def __init__(self, ptr_: int, nbytes: int, owner):
        self._owner = owner   # keeps the mapping alive as long as a tensor views it
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr_, False), "version": 2}


class _GrowableTensor:
    """VMMTensor / HostVMMTensor of the reference (src/vmm_tensor.cu:30-585): a 1-D buffer whose logical size is extended
    IN PLACE -- address space is reserved once (mi355_vmm_create), `extend` maps more physical memory at the tail
    (hipMemCreate + hipMemMap in HBM; mmap + hipHostRegister pinned host memory for the host flavour), `data()` is the
    current logical view and its data_ptr never changes.  `reserve_numel` bounds the growth (default: 16 x the initial
    size, at least 1 GiB)."""

    def __init__(self, numel: int, dtype: torch.dtype, device: int, host: bool, reserve_numel: Optional[int] = None):
        if numel <= 0:
            raise ValueError("numel must be positive")
        self._dtype, self._device, self._host = dtype, device, host
        self._eb = torch.empty((), dtype=dtype).element_size()
        self._logical = int(numel)
        reserve = max(int(reserve_numel) if reserve_numel else 16 * int(numel), int(numel)) * self._eb
        reserve = max(reserve, 1 << 30)
        h = c_p()
        check(lib().mi355_vmm_create(reserve, self._logical * self._eb, int(device), int(host), ctypes.byref(h)), "vmm_create")
        self._h = h

    def extend(self, new_total_logical_numel: int) -> None:
        n = int(new_total_logical_numel)
        if n <= self._logical:
            return
        check(lib().mi355_vmm_extend(self._h, n * self._eb), "vmm_extend")
        self._logical = n

    def data(self) -> torch.Tensor:
        base = lib().mi355_vmm_data(self._h)
        view = _CudaArrayView(int(base), self._logical * self._eb, self)
        if self._host:
            # pinned, registered host memory: torch sees it as a CPU tensor; kernels address it through the same pointer
            buf = (ctypes.c_uint8 * (self._logical * self._eb)).from_address(int(base))
            t = torch.frombuffer(buf, dtype=torch.uint8)
            t._vmm_owner = self
            return t.view(self._dtype)
        t = torch.as_tensor(view, device=torch.device("cuda", self._device))
        return t.view(self._dtype)

    def data_ptr(self) -> int:
        return int(lib().mi355_vmm_data(self._h))

    def logical_numel(self) -> int:
        return self._logical

    def allocated_numel(self) -> int:
        return int(lib().mi355_vmm_mapped_bytes(self._h)) // self._eb

    def allocated_bytes(self) -> int:
        return int(lib().mi355_vmm_mapped_bytes(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                lib().mi355_vmm_destroy(h)
            except Exception:
                pass
            self._h = None


class VMMTensor(_GrowableTensor):
    def __init__(self, numel: int, dtype: torch.dtype, device: int, reserve_numel: Optional[int] = None):
        super().__init__(numel, dtype, device, host=False, reserve_numel=reserve_numel)


class HostVMMTensor(_GrowableTensor):
    def __init__(self, numel: int, dtype: torch.dtype, device: int, reserve_numel: Optional[int] = None):
        super().__init__(numel, dtype, device, host=True, reserve_numel=reserve_numel)


def device_timestamp() -> int:
    """device_timestamp (torch_utils.cu:150): device clock ticks (host sync, as the reference)."""
    t = torch.empty(1, dtype=torch.int64, device="cuda")
    check(lib().mi355_device_timestamp(ptr(t), stream()), "device_timestamp")
    return int(t.item())


def bucketize_keys(keys, table_ids, table_bucket_offsets, num_buckets, bucket_capacity):
    """bucketize_keys (bucketize.cu:121-260): keys sorted by (bucket, key) for DEMB_DETERMINISM_MODE.
    Only used to order deterministic-mode insert waves; the ordering itself runs through torch's
    device sort (plumbing), the bucket id through the same hash as the table kernels."""
    n = keys.numel()
    if n == 0:
        e = torch.empty(0, dtype=torch.int64, device=keys.device)
        return e, e.clone(), e.clone()
    h = _hash63(keys)
    bb = table_bucket_offsets[table_ids]
    cap = (table_bucket_offsets[table_ids + 1] - bb) * bucket_capacity
    local = torch.where(cap > 0, torch.remainder(h, torch.clamp(cap, min=1)), torch.zeros_like(h))
    seg = bb + torch.div(local, bucket_capacity, rounding_mode="floor")
    k64 = keys.view(torch.int64)
    order = torch.argsort(k64, stable=True)
    order = order[torch.argsort(seg[order], stable=True)]
    seg_sorted = seg[order]
    change = torch.ones(n, dtype=torch.bool, device=keys.device)
    change[1:] = seg_sorted[1:] != seg_sorted[:-1]
    starts = torch.nonzero(change).flatten()
    offsets = torch.cat([starts, torch.tensor([n], device=keys.device, dtype=torch.int64)])
    return keys[order], offsets, order


def _hash63(keys: torch.Tensor) -> torch.Tensor:
    """fmix64 & INT64_MAX on int64 lanes with wrap-around arithmetic (types.cuh:123-131)."""
    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)

    k = keys.view(torch.int64).clone()
    k = k ^ lsr(k, 33)
    k = k * torch.tensor(-49064778989728563, dtype=torch.int64, device=k.device)  # 0xff51afd7ed558ccd
    k = k ^ lsr(k, 33)
    k = k * torch.tensor(-4265267296055464877, dtype=torch.int64, device=k.device)  # 0xc4ceb9fe1a85ec53
    k = k ^ lsr(k, 33)
    return k & 0x7FFFFFFFFFFFFFFF


def no_eviction_assign_scores(no_eviction_next_index_dev: torch.Tensor, table_ids: torch.Tensor) -> torch.Tensor:
    """no_eviction_scores.cu:19-29: per-table monotonically increasing row ids (order inside a call
    is arbitrary in the reference; here it is input order)."""
    n = table_ids.numel()
    out = torch.empty(n, dtype=torch.int64, device=table_ids.device)
    for t in range(no_eviction_next_index_dev.numel()):
        m = table_ids == t
        c = int(m.sum().item())
        if c:
            base = no_eviction_next_index_dev[t].clone()
            out[m] = base + torch.arange(c, device=out.device)
            no_eviction_next_index_dev[t] += c
    return out.view(torch.uint64)


# ------------------------------------------------------------------------------ index ops ----
_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    # fresh from the caching allocator each call: stream-ordered reuse is handled by torch
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def segmented_unique_cuda(keys, segmented_range, num_tables, input_frequencies=None):
    """segmented_unique_cuda (unique_op.cu:484-714) ->
    (num_uniques[1] (device view), unique_keys[N], output_indices i64[N], table_offsets i64[T+1], freq)."""
    if segmented_range.numel() != num_tables + 1:
        raise RuntimeError("segmented_range must have num_tables+1 elements")
    if segmented_range.dtype != torch.int64:
        raise RuntimeError("segmented_range must be int64")
    n = keys.numel()
    dev = keys.device
    count = input_frequencies is not None
    has_in = count and input_frequencies.numel() > 0
    if has_in and input_frequencies.numel() != n:
        raise RuntimeError("input_frequencies must have same length as keys")
    unique_keys = torch.empty_like(keys)
    out_idx = torch.empty(n, dtype=torch.int64, device=dev)
    table_offsets = torch.empty(num_tables + 1, dtype=torch.int64, device=dev)
    freq = torch.empty(n if count else 0, dtype=torch.int64, device=dev)
    wsb = lib().mi355_segmented_unique_workspace_bytes(n)
    ws = _workspace(wsb, dev)
    check(lib().mi355_segmented_unique(ptr(keys), n, ptr(segmented_range), num_tables,
                                       ptr(input_frequencies if has_in else None), int(count), ptr(unique_keys),
                                       ptr(out_idx), ptr(table_offsets), ptr(freq if count else None), ptr(ws),
                                       ws.numel(), stream()), "segmented_unique")
    return table_offsets[num_tables:], unique_keys, out_idx, table_offsets, freq


def segmented_unique_csr(keys, segmented_range, num_tables):
    """segmented_unique_cuda plus the CSR ingredients of the backward (extension): ->
    (unique_keys[N], output_indices i64[N], table_offsets i64[T+1], csr_cnt i32[N], csr_rank i32[N])."""
    n = keys.numel()
    dev = keys.device
    unique_keys = torch.empty_like(keys)
    out_idx = torch.empty(n, dtype=torch.int64, device=dev)
    table_offsets = torch.empty(num_tables + 1, dtype=torch.int64, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    rank = torch.empty(n, dtype=torch.int32, device=dev)
    ws = _workspace(lib().mi355_segmented_unique_workspace_bytes(n), dev)
    check(lib().mi355_segmented_unique_csr(ptr(keys), n, ptr(segmented_range), num_tables, None, 0, ptr(unique_keys),
                                           ptr(out_idx), ptr(table_offsets), None, ptr(cnt), ptr(rank), ptr(ws), ws.numel(),
                                           stream()), "segmented_unique_csr")
    return unique_keys, out_idx, table_offsets, cnt, rank


def group_by_unique_csr(csr_cnt, csr_rank, reverse_indices, num_unique_max, offsets=None, nu_dev=None, dim=0):
    """group_by_unique from the forward's counts / ranks: scan + scatter (no histogram, no atomics)."""
    n = reverse_indices.numel()
    dev = reverse_indices.device
    ptr_t = torch.empty(num_unique_max + 1, dtype=torch.int32, device=dev)
    csr = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ws = _workspace(lib().mi355_group_by_unique_csr_workspace_bytes(num_unique_max), dev)
    hot = _workspace(lib().mi355_hot_rows_workspace_bytes(n, dim), dev) if dim > 0 else None
    nb = offsets.numel() - 1 if offsets is not None else 0
    check(lib().mi355_group_by_unique_csr(ptr(csr_cnt), ptr(csr_rank), ptr(reverse_indices), n, ptr(offsets), nb,
                                          num_unique_max, ptr(nu_dev), ptr(ptr_t), ptr(csr), ptr(ws), ws.numel(), ptr(hot),
                                          hot.numel() if hot is not None else 0, dim, stream()), "group_by_unique_csr")
    if dim > 0:
        return ptr_t, csr, hot
    return ptr_t, csr


def expand_table_ids_cuda(offsets, num_elements=0, n_dev=None):
    out = torch.empty(num_elements, dtype=torch.int64, device=offsets.device)
    check(lib().mi355_expand_table_ids(ptr(offsets), offsets.numel() - 1, num_elements, ptr(n_dev), ptr(out), stream()),
          "expand_table_ids")
    return out


def compute_dedup_lengths_cuda(unique_offsets, table_offsets_in_feature, num_tables, local_batch_size, new_lengths_size):
    """compute_dedup_lengths_cuda (unique_op.cu:753-789) -> (new_lengths i64[n], new_offsets i64[n+1])."""
    if not unique_offsets.is_cuda:
        raise RuntimeError("unique_offsets must be on CUDA device")
    if not table_offsets_in_feature.is_cuda:
        raise RuntimeError("table_offsets_in_feature must be on CUDA device")
    dev = unique_offsets.device
    if new_lengths_size == 0:
        return torch.empty(0, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    nl = torch.empty(new_lengths_size, dtype=torch.int64, device=dev)
    no = torch.empty(new_lengths_size + 1, dtype=torch.int64, device=dev)
    check(lib().mi355_compute_dedup_lengths(ptr(unique_offsets), ptr(table_offsets_in_feature), num_tables,
                                            local_batch_size, new_lengths_size, ptr(nl), ptr(no), stream()),
          "compute_dedup_lengths")
    return nl, no


def segmented_sum_cuda(data, offsets):
    """segmented_sum_cuda (index_calculation.cu:38-75): int32 data, int64 offsets -> int64 [num_segments]."""
    if offsets.dtype != torch.int64:
        raise RuntimeError("offsets must be int64")
    if data.dtype != torch.int32:
        raise RuntimeError("data must be int32")
    ns = offsets.size(0) - 1
    if ns <= 0:
        raise RuntimeError("offsets size must be at least 2 (num_segments >= 1)")
    out = torch.empty(ns, dtype=torch.int64, device=data.device)
    check(lib().mi355_segmented_sum(ptr(data.contiguous()), ptr(offsets), ns, ptr(out), stream()), "segmented_sum")
    return out


def get_table_range(offsets, feature_offsets):
    """get_table_range (index_calculation.cu:93-127): range[t] = offsets[feature_offsets[t] * B]."""
    if not offsets.is_cuda:
        raise RuntimeError("Tensor <offsets> must be on CUDA device.")
    if not feature_offsets.is_cuda:
        raise RuntimeError("Tensor <feature_offsets> must be on CUDA device.")
    T = feature_offsets.numel() - 1
    out = torch.empty_like(feature_offsets)
    check(lib().mi355_get_table_range(ptr(offsets), ptr(feature_offsets), T, offsets.numel() - 1, ptr(out), stream()),
          "get_table_range")
    return out


def flagged_compact(flags: torch.Tensor, inputs: List[Optional[torch.Tensor]]):
    """flagged_compact (index_calculation.cu:129-232) -> (count:int, indices, [tensors]).
    The reference returns the count as a host int (one sync); so does this wrapper.  The sync-free
    form is `flagged_compact_async` below."""
    cnt, idx, outs = flagged_compact_async(flags, inputs)
    c = int(cnt.item())
    return c, idx[:c], [None if o is None else o[:c] for o in outs]


def flagged_compact_async(flags, tensors, n_dev=None):
    n = flags.numel()
    dev = flags.device
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    present = [t for t in tensors if t is not None]
    for t in present:
        if t.element_size() != 8:
            raise RuntimeError("flagged_compact carries 8-byte arrays only")
    outs_present = [torch.empty_like(t) for t in present]
    arr_t = c_p * max(len(present), 1)
    ins = arr_t(*[t.data_ptr() for t in present]) if present else arr_t(0)
    outs = arr_t(*[t.data_ptr() for t in outs_present]) if present else arr_t(0)
    ws = _workspace(lib().mi355_flagged_compact_workspace_bytes(n), dev)
    check(lib().mi355_flagged_compact(ptr(_u8(flags)), n, ptr(n_dev), ptr(cnt), ptr(idx), len(present), ins, outs,
                                      ptr(ws), ws.numel(), stream()), "flagged_compact")
    it = iter(outs_present)
    return cnt, idx, [None if t is None else next(it) for t in tensors]


def block_bucketize_sparse_features(lengths, indices, bucketize_pos, sequence, dist_type_per_feature=None,
                                    block_sizes=None, my_size=None, weights=None, batch_size_per_feature=None, max_B=-1,
                                    block_bucketize_pos=None):
    """block_bucketize_sparse_features (sparse_block_bucketize_features.cu:366-830) ->
    (new_lengths, new_indices, new_weights, new_pos, unbucketize_permute)."""
    want_perm = sequence
    sequence = sequence or bucketize_pos      # (the positions travel with the permutation)
    FB = lengths.numel()
    F = block_sizes.numel()
    dev = indices.device
    # variable batch size per feature (the reference builds length_to_feature_idx from it, :194-211 / :430-470): the bags of
    # feature f are [starts[f], starts[f + 1])
    fstart = None
    if batch_size_per_feature is not None:
        bspf = torch.as_tensor(batch_size_per_feature, device=dev).to(torch.int64).view(-1)
        if bspf.numel() != F:
            raise RuntimeError("batch_size_per_feature must have one entry per feature")
        # (round-4 advisor) the bags past the last start would be attributed to feature F - 1, a short sum would misroute them: when
        # the sizes are host values (a list / tuple / CPU tensor, as TorchRec passes them) the sum is checked here, without a sync
        if not (isinstance(batch_size_per_feature, torch.Tensor) and batch_size_per_feature.is_cuda):
            total = int(sum(int(x) for x in (batch_size_per_feature.tolist() if isinstance(batch_size_per_feature, torch.Tensor)
                                            else batch_size_per_feature)))
            if total != FB:
                raise RuntimeError(f"sum(batch_size_per_feature) = {total} does not match lengths.numel() = {FB}")
        fstart = torch.zeros(F + 1, dtype=torch.int64, device=dev)
        torch.cumsum(bspf, 0, out=fstart[1:])
        B = 0
    else:
        B = FB // F
    # uneven shard boundaries: one sorted tensor per feature (my_size + 1 boundaries as TorchRec builds them)
    pos_cat = pos_off = None
    if block_bucketize_pos is not None:
        if len(block_bucketize_pos) != F:
            raise RuntimeError("block_bucketize_pos must have one tensor per feature")
        pos_cat = torch.cat([torch.as_tensor(x, device=dev).to(torch.int64).view(-1) for x in block_bucketize_pos]).contiguous()
        sizes = [int(torch.as_tensor(x).numel()) for x in block_bucketize_pos]
        pos_off = torch.tensor([sum(sizes[:i]) for i in range(F + 1)], dtype=torch.int64, device=dev)
    offsets = torch.zeros(FB + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lengths.to(torch.int64), 0, out=offsets[1:])
    new_lengths = torch.empty(my_size * FB, dtype=torch.int64, device=dev)
    new_offsets = torch.empty(my_size * FB + 1, dtype=torch.int64, device=dev)
    new_indices = torch.empty_like(indices)
    perm = torch.empty(indices.numel(), dtype=torch.int64, device=dev) if sequence else None
    new_w = torch.empty_like(weights) if weights is not None else None
    dist = dist_type_per_feature.to(torch.int32) if dist_type_per_feature is not None else None
    check(lib().mi355_block_bucketize_ex(my_size, FB, B, ptr(offsets), ptr(indices), ptr(block_sizes.to(torch.int64)),
                                         ptr(dist), ptr(weights), ptr(new_lengths), ptr(new_offsets), ptr(new_indices),
                                         ptr(new_w), ptr(perm), ptr(fstart), F, ptr(pos_cat), ptr(pos_off), stream()), "block_bucketize")
    new_pos = None
    if bucketize_pos:
        # bucketize_pos (sparse_block_bucketize_features.cu:366-830, `new_pos`): the position every value had inside its ORIGINAL
        # bag, carried to the value's place in the bucketized order (what TorchRec's position-weighted feature processors read)
        n = indices.numel()
        bag = torch.repeat_interleave(torch.arange(FB, device=dev), lengths.to(torch.int64), output_size=n)
        new_pos = torch.empty(n, dtype=indices.dtype, device=dev)
        new_pos[perm] = (torch.arange(n, device=dev) - offsets[bag]).to(indices.dtype)
    return new_lengths.to(lengths.dtype), new_indices, new_w, new_pos, (perm if want_perm else None)


# ------------------------------------------------------------------------------ value ops ----
def _aligned(*vals) -> int:
    return int(all(v % 4 == 0 for v in vals))


def gather_embedding(input: torch.Tensor, output: torch.Tensor, index: torch.Tensor):
    if output.size(0) != index.numel():
        raise RuntimeError("Number rows of `output` must match with `index`.")
    if output.size(1) != input.size(1):
        raise RuntimeError("Number cols of `output` must match with `input`.")
    D = output.size(1)
    al = _aligned(D, input.stride(0), output.stride(0)) and input.data_ptr() % 16 == 0 and output.data_ptr() % 16 == 0
    check(lib().mi355_gather_rows(ptr(input), input.stride(0), None, dt(input), ptr(index), index.numel(), None, D,
                                  ptr(output), output.stride(0), dt(output), int(al), stream()), "gather_embedding")


def gather_embedding_pooled(input, output, index, offsets, combiner, total_D, batch_size, D_offsets=None, max_D=0,
                            row_addr=None, src_dtype=None):
    """gather_embedding_pooled (dynamic_emb_op.cu:106-133).  `row_addr` (extension): pool straight from
    the table rows instead of the dense `input`."""
    num_bags = offsets.numel() - 1
    if D_offsets is not None and D_offsets.dtype != torch.int32:
        raise RuntimeError(f"D_offsets must be int32, got {D_offsets.dtype}")
    dim = max_D if D_offsets is not None else (input.size(1) if input is not None else max_D)
    stride = input.stride(0) if input is not None else 0
    sdt = dt(input) if input is not None else dt(src_dtype)
    al = _aligned(dim, total_D, stride) and output.data_ptr() % 16 == 0
    if D_offsets is not None:
        al = al and bool((D_offsets % 4 == 0).all().item()) if D_offsets.numel() < 4096 else al
    if input is not None:
        al = al and input.data_ptr() % 16 == 0
    check(lib().mi355_gather_pooled(ptr(input), stride, ptr(row_addr), sdt, ptr(index), index.numel(), ptr(offsets), num_bags,
                                    batch_size, int(combiner), dim, ptr(D_offsets), total_D, ptr(output), dt(output),
                                    int(al), stream()), "gather_embedding_pooled")


def _flat_copy(is_load, region, table_ptrs, indices, table_ids, scalar_table_id, dense, table_value_dims,
               table_emb_dims, max_emb_dim):
    if dense.dim() != 2:
        raise RuntimeError("output must be 2-D" if is_load else "input must be 2-D")
    if dense.size(0) != indices.numel():
        raise RuntimeError("size(0) must match indices.size(0)")
    check(lib().mi355_flat_table_copy(int(is_load), region, indices.numel(), None, ptr(dense), dense.size(1),
                                      dense.stride(0), dt(dense), ptr(indices), ptr(table_ids), scalar_table_id,
                                      ptr(table_ptrs), ptr(table_value_dims), ptr(table_emb_dims), max_emb_dim,
                                      stream()), "flat_table_copy")


def load_from_flat_table_contiguous(table_ptrs, indices, table_id, output, table_value_dims, table_emb_dims,
                                    max_emb_dim, all_dims_vec4):
    _flat_copy(True, 0, table_ptrs, indices, None, table_id, output, table_value_dims, table_emb_dims, max_emb_dim)


def load_from_flat_table_emb(table_ptrs, indices, table_ids, output, table_value_dims, table_emb_dims, max_emb_dim,
                             all_dims_vec4):
    _flat_copy(True, 1, table_ptrs, indices, table_ids, 0, output, table_value_dims, table_emb_dims, max_emb_dim)


def load_from_flat_table_value(table_ptrs, indices, table_ids, output, table_value_dims, table_emb_dims, max_emb_dim,
                               all_dims_vec4):
    _flat_copy(True, 2, table_ptrs, indices, table_ids, 0, output, table_value_dims, table_emb_dims, max_emb_dim)


def store_to_flat_table_contiguous(table_ptrs, indices, table_id, input, table_value_dims, table_emb_dims,
                                   max_emb_dim, all_dims_vec4):
    _flat_copy(False, 0, table_ptrs, indices, None, table_id, input, table_value_dims, table_emb_dims, max_emb_dim)


def store_to_flat_table_value(table_ptrs, indices, table_ids, input, table_value_dims, table_emb_dims, max_emb_dim,
                              all_dims_vec4):
    _flat_copy(False, 2, table_ptrs, indices, table_ids, 0, input, table_value_dims, table_emb_dims, max_emb_dim)


def row_addresses(slots, table_ids, table_ptrs, table_value_dims, elem_bytes, n_dev=None):
    out = torch.empty(slots.numel(), dtype=torch.int64, device=slots.device)
    check(lib().mi355_row_addresses(slots.numel(), ptr(n_dev), ptr(slots), ptr(table_ids), ptr(table_ptrs),
                                    ptr(table_value_dims), elem_bytes, ptr(out), stream()), "row_addresses")
    return out


def select_insert_failed_values(indices, input_values, evicted_values):
    """select_insert_failed_values (dynamic_emb_op.cu:686-799): evicted_values[i] = input_values[-(idx+1)]
    for the Busy entries (idx < 0) of an insert_and_evict call."""
    src = torch.where(indices < 0, -(indices + 1), torch.full_like(indices, -1))
    D = evicted_values.size(1)
    al = _aligned(D, input_values.stride(0), evicted_values.stride(0))
    check(lib().mi355_gather_rows(ptr(input_values), input_values.stride(0), None, dt(input_values), ptr(src),
                                  src.numel(), None, D, ptr(evicted_values), evicted_values.stride(0),
                                  dt(evicted_values), int(al), stream()), "select_insert_failed_values")


# ----------------------------------------------------------------------------- initializers ----
class CurandStateContext:
    """Stands in for CurandStateContext (initializer.cu:186-195).  The MI355X build uses a counter
    based generator keyed by (seed, row key, element): no per-thread state buffer is needed."""

    def __init__(self, seed: int = 1234):
        self.seed = int(seed)


def _init(mode, buffer, indices, keys, p, ctx=None):
    seed = ctx.seed if ctx is not None else 0
    n = indices.numel() if indices is not None else buffer.size(0)
    check(lib().mi355_init_rows(mode, c_f(p[0]), c_f(p[1]), c_f(p[2]), c_f(p[3]), c_u64(seed), c_f(0.0), n, None,
                                ptr(keys if keys is not None else torch.arange(buffer.size(0), device=buffer.device)),
                                ptr(indices), None, ptr(buffer), buffer.stride(0), dt(buffer), buffer.size(1),
                                buffer.size(1), None, None, None, None, None, stream()), "init_rows")


def uniform_init(buffer, indices, curand_state_context, lower, upper, keys=None):
    _init(0, buffer, indices, keys, (lower, upper, 0, 0), curand_state_context)


def normal_init(buffer, indices, curand_state_context, mean, std_dev, keys=None):
    _init(1, buffer, indices, keys, (mean, std_dev, 0, 0), curand_state_context)


def truncated_normal_init(buffer, indices, curand_state_context, mean, std_dev, lower, upper, keys=None):
    _init(2, buffer, indices, keys, (mean, std_dev, lower, upper), curand_state_context)


def const_init(buffer, indices, value):
    _init(3, buffer, indices, None, (value, 0, 0, 0))


def debug_init(buffer, indices, keys):
    _init(4, buffer, indices, keys, (0, 0, 0, 0))


# --------------------------------------------------------------------------------- backward ----
def group_by_unique(reverse_indices, num_unique_max, offsets=None, nu_dev=None, dim=0):
    """CSR (ptr int32[Nu+1], csr_src int32[Nt]) of the batch keyed by unique row, plus (dim > 0) the
    hot-row task list consumed by backward_fused."""
    n = reverse_indices.numel()
    dev = reverse_indices.device
    ptr_t = torch.empty(num_unique_max + 1, dtype=torch.int32, device=dev)
    csr = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ws = _workspace(lib().mi355_group_by_unique_workspace_bytes(n, num_unique_max), dev)
    hot = _workspace(lib().mi355_hot_rows_workspace_bytes(n, dim), dev) if dim > 0 else None
    nb = offsets.numel() - 1 if offsets is not None else 0
    check(lib().mi355_group_by_unique(ptr(reverse_indices), n, ptr(offsets), nb, num_unique_max, ptr(nu_dev),
                                      ptr(ptr_t), ptr(csr), ptr(ws), ws.numel(), ptr(hot),
                                      hot.numel() if hot is not None else 0, dim, stream()), "group_by_unique")
    if dim > 0:
        return ptr_t, csr, hot
    return ptr_t, csr


def backward_fused(ptr_t, csr, num_keys, num_unique_max, grads, batch_size, dim, combiner, offsets=None,
                   D_offsets=None, row_addr=None, weight_dtype=torch.float32, opt_kind=0, lr=0.0, beta1=0.9,
                   beta2=0.999, eps=1e-8, weight_decay=0.0, iter_num=1, state_offset=None, round_grad=True, out=None,
                   nu_dev=None, hot=None):
    """`hot`: the hot-row task list from group_by_unique(..., dim=dim); None = every row on the serial path."""
    gs = grads.stride(0)
    al = _aligned(dim, gs) and grads.data_ptr() % 16 == 0
    if D_offsets is not None:
        al = al and bool((D_offsets % 4 == 0).all().item())
    if out is not None:
        al = al and out.stride(0) % 4 == 0
    so = -1 if state_offset is None else state_offset
    if opt_kind != 0 and state_offset is not None:
        al = al and so % 4 == 0
    check(lib().mi355_backward_fused(ptr(ptr_t), ptr(csr), num_keys, num_unique_max, ptr(nu_dev), ptr(grads), gs,
                                     dt(grads), ptr(offsets), ptr(D_offsets), batch_size, dim, combiner,
                                     ptr(row_addr), dt(weight_dtype), opt_kind, c_f(lr), c_f(beta1), c_f(beta2),
                                     c_f(eps), c_f(weight_decay), iter_num, so, int(round_grad), ptr(out),
                                     out.stride(0) if out is not None else 0, int(al), ptr(hot),
                                     hot.numel() if hot is not None else 0, stream()), "backward_fused")


def reduce_grads(reverse_indices, grads, num_unique, batch_size, out_dim, offsets=None, D_offsets=None, combiner=-1,
                 total_D=0, out_dtype=None):
    """reduce_grads (dynamic_emb_op.cu:159-285) -> unique_grads [num_unique, out_dim] in the grad dtype
    (`out_dtype`, an extension, keeps the fp32 sums -- used by the sharded backward)."""
    out_dtype = out_dtype or grads.dtype
    unique_grads = torch.empty(num_unique, out_dim, dtype=out_dtype, device=grads.device)
    n = reverse_indices.numel()
    if n == 0 or batch_size == 0 or num_unique == 0:
        return unique_grads
    pooled = offsets is not None
    ptr_t, csr, hot = group_by_unique(reverse_indices, num_unique, offsets if pooled else None, dim=out_dim)
    backward_fused(ptr_t, csr, n, num_unique, grads.contiguous(), batch_size, out_dim,
                   combiner if pooled else -1, offsets if pooled else None, D_offsets if pooled else None,
                   weight_dtype=out_dtype, opt_kind=0, out=unique_grads, round_grad=False, hot=hot)
    return unique_grads


def _opt_flat(kind, grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, max_emb_dim, hp,
              weight_dtype):
    # rows addressed through (table_ptrs, table_ids, indices): optimizer.cu:34-75
    eb = _BYTES[weight_dtype]
    addr = row_addresses(indices, table_ids, table_ptrs, table_value_dims, eb)
    D = grads.size(1)
    al = _aligned(D, grads.stride(0))
    check(lib().mi355_optimizer_update(kind, ptr(grads), grads.stride(0), dt(grads), grads.size(0), None, ptr(addr),
                                       None, 0, dt(weight_dtype), D, D, c_f(hp.get("lr", 0.0)), c_f(hp.get("beta1", 0.9)),
                                       c_f(hp.get("beta2", 0.999)), c_f(hp.get("eps", 1e-8)),
                                       c_f(hp.get("weight_decay", 0.0)), int(hp.get("iter_num", 1)), int(al), stream()),
          "optimizer_update")


_DT_FROM_ENUM = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


def _table_dtype(d):
    """table_dtype arrives as a DynamicEmbDataType value from the reference's optimizer.py (torch_to_dyn_emb(..).value)
    or as a torch dtype."""
    if isinstance(d, torch.dtype):
        return d
    return _DT_FROM_ENUM[int(d)]


# Positional orders are the pybind ones of src/optimizer.cu:415-447 (the reference calls them positionally).
def sgd_update_for_flat_table(grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, max_emb_dim,
                              all_dims_vec4, lr, table_dtype=torch.float32):
    _opt_flat(1, grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, max_emb_dim, dict(lr=lr),
              _table_dtype(table_dtype))


def adam_update_for_flat_table(grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, lr, beta1, beta2,
                               eps, weight_decay, iter_num, max_emb_dim, all_dims_vec4, table_dtype=torch.float32):
    _opt_flat(2, grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, max_emb_dim,
              dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, iter_num=iter_num),
              _table_dtype(table_dtype))


def adagrad_update_for_flat_table(grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, lr, eps,
                                  max_emb_dim, all_dims_vec4, table_dtype=torch.float32):
    _opt_flat(3, grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, max_emb_dim,
              dict(lr=lr, eps=eps), _table_dtype(table_dtype))


def rowwise_adagrad_for_flat_table(grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, lr, eps,
                                   max_emb_dim, all_dims_vec4, table_dtype=torch.float32):
    _opt_flat(4, grads, indices, table_ptrs, table_ids, table_value_dims, table_emb_dims, max_emb_dim,
              dict(lr=lr, eps=eps), _table_dtype(table_dtype))


rowwise_adagrad_update_for_flat_table = rowwise_adagrad_for_flat_table


def _opt_padded(kind, grads, values, emb_dim, state_offset, hp, table_ids=None, table_emb_dims=None):
    """{sgd,adam,adagrad,rowwise_adagrad}_update_for_padded_buffer (optimizer.cu:242-413): row u of `values` is
    table_emb_dims[table_ids[u]] wide (all `emb_dim` wide when the tables agree), its state starts at `state_offset` = the
    widest embedding."""
    D = emb_dim
    al = _aligned(D, grads.stride(0), values.stride(0), state_offset)
    mixed = table_ids is not None and table_emb_dims is not None and table_emb_dims.numel() > 1
    if mixed and al and table_emb_dims.numel() <= 4096:
        al = int(bool((table_emb_dims % 4 == 0).all().item()))
    check(lib().mi355_optimizer_update_tables(kind, ptr(grads), grads.stride(0), dt(grads), grads.size(0), None, None,
                                              ptr(values), values.stride(0), dt(values), D, state_offset,
                                              c_f(hp.get("lr", 0.0)), c_f(hp.get("beta1", 0.9)), c_f(hp.get("beta2", 0.999)),
                                              c_f(hp.get("eps", 1e-8)), c_f(hp.get("weight_decay", 0.0)),
                                              int(hp.get("iter_num", 1)), int(al),
                                              ptr(table_ids.to(torch.int64).contiguous()) if mixed else None,
                                              ptr(table_emb_dims.to(torch.int64).contiguous()) if mixed else None, stream()),
          "optimizer_update_padded")


def sgd_update_for_padded_buffer(grads, values, table_ids, table_emb_dims, emb_dim, value_dim, all_dims_vec4, lr):
    _opt_padded(1, grads, values, emb_dim, emb_dim, dict(lr=lr), table_ids, table_emb_dims)


def adam_update_for_padded_buffer(grads, values, table_ids, table_emb_dims, emb_dim, value_dim, all_dims_vec4,
                                  lr, beta1, beta2, eps, weight_decay, iter_num):
    _opt_padded(2, grads, values, emb_dim, emb_dim,
                dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, iter_num=iter_num), table_ids, table_emb_dims)


def adagrad_update_for_padded_buffer(grads, values, table_ids, table_emb_dims, emb_dim, value_dim, all_dims_vec4,
                                     lr, eps):
    _opt_padded(3, grads, values, emb_dim, emb_dim, dict(lr=lr, eps=eps), table_ids, table_emb_dims)


def rowwise_adagrad_for_padded_buffer(grads, values, table_ids, table_emb_dims, emb_dim, value_dim, all_dims_vec4,
                                      lr, eps):
    _opt_padded(4, grads, values, emb_dim, emb_dim, dict(lr=lr, eps=eps), table_ids, table_emb_dims)


rowwise_adagrad_update_for_padded_buffer = rowwise_adagrad_for_padded_buffer


def init_rows(mode, params, seed, state_init, keys, row_addr, dtype, emb_dim, value_dim, results=None, skip=None,
              n_dev=None, table_ids=None, table_emb_dims=None, table_value_dims=None):
    """first-touch initialisation of table rows IN PLACE (fuses initializer + store_to_flat of
    batched_dynamicemb_function.py:648-676)."""
    p = list(params) + [0.0] * (4 - len(params))
    check(lib().mi355_init_rows(int(mode), c_f(p[0]), c_f(p[1]), c_f(p[2]), c_f(p[3]), c_u64(seed), c_f(state_init),
                                keys.numel(), ptr(n_dev), ptr(keys), None, ptr(row_addr), None, 0, dt(dtype), emb_dim,
                                value_dim, ptr(_u8(results)), ptr(_u8(skip)), ptr(table_ids), ptr(table_emb_dims),
                                ptr(table_value_dims), stream()), "init_rows")
