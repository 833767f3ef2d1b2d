"""ctypes loader of librecsys_amd.so (the C ABI of include/recsys_amd.h).

PyTorch is plumbing only: tensors provide device memory and the current HIP stream; every
compute call goes through the C ABI with raw pointers.  There is NO CPU or eager fallback: if
the library is missing, or a call is made without a GPU, this raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_LIB", os.path.join(_HERE, "lib", "librecsys_amd.so"))

c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64
c_int = ctypes.c_int
c_f = ctypes.c_float
c_p = ctypes.c_void_p

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: DT_F16}


class NativeError(RuntimeError):
    pass


_lib = None

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGS = {
    "mi355_table_init": [c_p, c_i64, c_i64, c_i64, c_p],
    "mi355_table_lookup": [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_int, c_u64, c_p, c_p, c_p, c_p],
    "mi355_table_insert": [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_int, c_u64, c_p, c_p, c_p,
                           c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "mi355_table_erase": [c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p],
    "mi355_table_update_counter": [c_p, c_i64, c_p, c_i64, c_p, ctypes.c_int32, c_p, c_p, c_i64, c_p],
    "mi355_table_lookup_overflow": [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_int, c_u64, c_p, c_i64, c_p, c_p,
                                    c_p, c_p, c_p],
    "mi355_table_insert_overflow_workspace_bytes": [c_i64],
    "mi355_table_insert_overflow": [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_int, c_u64, c_p,
                                    c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_table_update_counter_overflow": [c_p, c_i64, c_p, c_i64, c_p, ctypes.c_int32, c_p, c_p, c_i64, c_i64, c_p,
                                            c_i64, c_p],
    "mi355_device_timestamp": [c_p, c_p],
    "mi355_table_export_batch": [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_u64, c_i64, c_i64, c_p, c_p, c_p, c_p,
                                 c_p, c_i64, c_p],
    "mi355_table_export_batch_workspace_bytes": [c_i64],
    "mi355_table_count_matched": [c_p, c_i64, c_i64, c_i64, c_u64, c_i64, c_i64, c_i64, c_p, c_p],
    "mi355_table_score_blocks": [c_int, c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p],
    "mi355_compute_dedup_lengths": [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p],
    "mi355_segmented_sum": [c_p, c_p, c_i64, c_p, c_p],
    "mi355_segmented_unique": [c_p, c_i64, c_p, c_i64, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_segmented_unique_csr": [c_p, c_i64, c_p, c_i64, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_group_by_unique_csr": [c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_p],
    "mi355_group_by_unique_csr_workspace_bytes": [c_i64],
    "mi355_expand_table_ids": [c_p, c_i64, c_i64, c_p, c_p, c_p],
    "mi355_get_table_range": [c_p, c_p, c_i64, c_i64, c_p, c_p],
    "mi355_flagged_compact": [c_p, c_i64, c_p, c_p, c_p, c_int, c_p, c_p, c_p, c_i64, c_p],
    "mi355_group_by_unique": [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_p],
    "mi355_hot_rows_workspace_bytes": [c_i64, c_i64],
    "mi355_permute_lengths": [c_i64, c_i64, c_i64, c_p, c_p, c_p],
    "mi355_permute_bags": [c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p],
    "mi355_sum_chunks": [c_p, c_i64, c_i64, c_p, c_int, c_p],
    "mi355_sum_chunks_typed": [c_p, c_int, c_i64, c_i64, c_p, c_int, c_p],
    "mi355_exclusive_offsets": [c_p, c_i64, c_p, c_p],
    "mi355_peer_splits": [c_p, c_p, c_i64, c_i64, c_p, c_p],
    "mi355_chunk_bags": [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p],
    "mi355_block_bucketize": [c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "mi355_block_bucketize_ex": [c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_p],
    "mi355_gather_pooled": [c_p, c_i64, c_p, c_int, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_i64, c_p, c_i64, c_p, c_int,
                            c_int, c_p],
    "mi355_gather_rows": [c_p, c_i64, c_p, c_int, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_int, c_int, c_p],
    "mi355_flat_table_copy": [c_int, c_int, c_i64, c_p, c_p, c_i64, c_i64, c_int, c_p, c_p, c_i64, c_p, c_p, c_p,
                              c_i64, c_p],
    "mi355_row_addresses": [c_i64, c_p, c_p, c_p, c_p, c_p, c_int, c_p, c_p],
    "mi355_init_rows": [c_int, c_f, c_f, c_f, c_f, c_u64, c_f, c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_i64,
                        c_i64, c_p, c_p, c_p, c_p, c_p, c_p],
    "mi355_demb_forward": [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64,  # table
                           c_p, c_p, c_p, c_int, c_i64, c_i64,  # values
                           c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i64,  # batch
                           c_int, c_int, c_p, c_int, c_p, c_u64, c_int,  # policies
                           c_int, c_f, c_f, c_f, c_f, c_u64, c_f,  # initializer
                           c_int, c_p, c_i64, c_p, c_int, c_int,  # output
                           c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,  # persisted (+ freq, csr_cnt, csr_rank)
                           c_p, c_i64,  # backward workspace (early CSR)
                           c_p,  # join token (out)
                           c_p, c_i64, c_p],
    "mi355_demb_forward_fused": [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_i64,  # table + aux
                                 c_p, c_p, c_p, c_int, c_i64, c_i64,  # values
                                 c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i64,  # batch
                                 c_int, c_int, c_int, c_u64, c_int, c_u64, c_int,  # policies
                                 c_int, c_f, c_f, c_f, c_f, c_u64, c_f,  # initializer
                                 c_int, c_p, c_i64, c_p, c_int, c_int,  # output
                                 c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,  # persisted
                                 c_p, c_i64, c_int, c_p,  # backward workspace, side stream, join token
                                 c_p, c_i64, c_p],
    "mi355_demb_forward_fused_rerun": [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_i64,  # table + aux
                                 c_p, c_p, c_p, c_int, c_i64, c_i64,  # values
                                 c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i64,  # batch
                                 c_int, c_int, c_int, c_u64, c_int, c_u64, c_int,  # policies
                                 c_int, c_f, c_f, c_f, c_f, c_u64, c_f,  # initializer
                                 c_int, c_p, c_i64, c_p, c_int, c_int,  # output
                                 c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,  # persisted
                                 c_p, c_i64, c_int, c_int,  # backward workspace, side stream, epoch
                                 c_p, c_i64, c_p],
    "mi355_demb_plan_create": [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_i64,  # table + aux
                               c_p, c_p, c_p, c_int, c_i64, c_i64, c_p, c_i64,  # values, feature offsets, tables
                               c_int, c_int, c_int, c_int,  # policies, pin
                               c_int, c_f, c_f, c_f, c_f, c_u64, c_f,  # initializer
                               c_int, c_p, c_i64, c_int, c_int,  # output
                               c_int, c_f, c_f, c_f, c_f],  # optimizer
    "mi355_demb_plan_destroy": [c_p],
    "mi355_demb_step_layout": [c_i64, c_i64, c_i64, c_int, c_p],
    "mi355_demb_plan_step_bytes": [c_p, c_i64],
    "mi355_demb_plan_forward": [c_p, c_p, c_i64, c_p, c_i64, c_i64, c_u64, c_u64, c_p, c_p, c_i64, c_p, c_p],
    "mi355_demb_plan_stage": [c_p, c_int, c_u64, c_p, c_i64, c_p, c_i64, c_i64, c_u64, c_u64, c_p, c_p, c_i64, c_p, c_p, c_int, c_p],
    "mi355_demb_plan_backward": [c_p, c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_int, c_int,
                                 c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_p],
    "mi355_demb_plan_rerun": [c_p, c_p, c_i64, c_p, c_i64, c_i64, c_u64, c_u64, c_p, c_i64, c_int, c_p],
    "mi355_demb_fused_step_flooded": [c_int, c_int],
    "mi355_demb_aux_numel": [c_i64, c_i64],
    "mi355_demb_forward_fused_workspace_bytes": [c_i64, c_i64],
    "mi355_demb_forward_fused_partitions": [c_i64, c_i64, c_i64],
    "mi355_side_join": [c_int, c_p],
    "mi355_demb_fused_materialize": [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p],
    "mi355_profile_kernels": [c_int],
    "mi355_profile_ms": [c_int],
    "mi355_demb_backward": [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_int, c_p, c_i64, c_int, c_p,
                            c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_int, c_int, c_p, c_i64, c_p, c_p,
                            c_p, c_i64, c_int, c_p, c_p, c_int, c_p, c_i64, c_p],
    "mi355_demb_forward_workspace_bytes": [c_i64, c_i64],
    "mi355_early_csr_stream": [],
    "mi355_demb_backward_workspace_bytes": [c_i64, c_i64],
    "mi355_backward_fused": [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_int, c_p, c_p, c_i64, c_i64, c_int, c_p,
                             c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_int, c_p, c_i64, c_int, c_p,
                             c_i64, c_p],
    "mi355_optimizer_update": [c_int, c_p, c_i64, c_int, c_i64, c_p, c_p, c_p, c_i64, c_int, c_i64, c_i64, c_f, c_f,
                               c_f, c_f, c_f, c_i64, c_int, c_p],
    "mi355_optimizer_update_tables": [c_int, c_p, c_i64, c_int, c_i64, c_p, c_p, c_p, c_i64, c_int, c_i64, c_i64, c_f, c_f,
                                      c_f, c_f, c_f, c_i64, c_int, c_p, c_p, c_p],
    "mi355_segmented_unique_workspace_bytes": [c_i64],
    "mi355_flagged_compact_workspace_bytes": [c_i64],
    "mi355_group_by_unique_workspace_bytes": [c_i64, c_i64],
    "mi355_backward_workspace_bytes": [c_i64, c_i64],
    "mi355_vmm_create": [c_i64, c_i64, c_int, c_int, c_p],
    "mi355_vmm_extend": [c_p, c_i64],
    "mi355_vmm_data": [c_p],
    "mi355_vmm_mapped_bytes": [c_p],
    "mi355_vmm_reserved_bytes": [c_p],
    "mi355_vmm_destroy": [c_p],
    "mi355_sum_chunks_self": [c_p, c_int, c_i64, c_i64, c_p, c_i64, c_p, c_int, c_p],
    "mi355_rw_load_rccl": [ctypes.c_char_p],
    "mi355_rw_unique_id": [c_p, c_i64],
    "mi355_rw_create": [c_p, c_p, c_int, c_int, c_p],
    "mi355_rw_destroy": [c_p],
    "mi355_rw_abort": [c_p],
    "mi355_rw_keys_ready": [c_p, c_int],
    "mi355_rw_input_cancel": [c_p, c_int],
    "mi355_rw_input_begin": [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "mi355_rw_input_counts": [c_p, c_int, c_p, c_p, c_p],
    "mi355_rw_input_counts_ready": [c_p, c_int],
    "mi355_rw_input_keys": [c_p, c_int, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "mi355_rw_wait_keys": [c_p, c_int, c_p],
    "mi355_rw_output_pooled": [c_p, c_p, c_p, c_i64, c_int, c_p, c_int, c_p],
    "mi355_rw_allgather": [c_p, c_p, c_p, c_i64, c_p],
    "mi355_rw_alltoallv": [c_p, c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_abi_version": [],
    "mi355_last_error": [],
}
_RESTYPES = {
    "mi355_segmented_unique_workspace_bytes": c_i64,
    "mi355_group_by_unique_csr_workspace_bytes": c_i64,
    "mi355_table_export_batch_workspace_bytes": c_i64,
    "mi355_table_insert_overflow_workspace_bytes": c_i64,
    "mi355_flagged_compact_workspace_bytes": c_i64,
    "mi355_group_by_unique_workspace_bytes": c_i64,
    "mi355_backward_workspace_bytes": c_i64,
    "mi355_hot_rows_workspace_bytes": c_i64,
    "mi355_demb_forward_workspace_bytes": c_i64,
    "mi355_demb_backward_workspace_bytes": c_i64,
    "mi355_early_csr_stream": c_p,
    "mi355_demb_plan_create": c_p,
    "mi355_demb_plan_destroy": None,
    "mi355_demb_step_layout": None,
    "mi355_demb_plan_step_bytes": c_i64,
    "mi355_vmm_data": c_p,
    "mi355_vmm_mapped_bytes": c_i64,
    "mi355_vmm_reserved_bytes": c_i64,
    "mi355_profile_ms": c_f,
    "mi355_demb_aux_numel": c_i64,
    "mi355_demb_forward_fused_workspace_bytes": c_i64,
    "mi355_last_error": ctypes.c_char_p,
}
_OPTIONAL_SIGS = {}  # filled by optional modules (e.g. hstu) before first load


def register_signatures(sigs, restypes=None):
    _OPTIONAL_SIGS.update(sigs)
    if restypes:
        _RESTYPES.update(restypes)
    if _lib is not None:
        _bind(_lib, sigs)


def signature_of(name):
    """argument types registered for `name` (for twins that share a signature, e.g. the fp16 entry points of the attention)"""
    return (_OPTIONAL_SIGS.get(name) or _SIGS[name])


def _bind(lib, sigs):
    for name, args in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = args
        fn.restype = _RESTYPES[name] if name in _RESTYPES else c_int


def exported_symbols():
    return sorted(list(_SIGS) + list(_OPTIONAL_SIGS))


def lib():
    """Loads the HIP library.  Raises NativeError if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python recsys-examples_amd/build.py` "
                "(or __graft_entry__.build()); this package has no CPU/eager fallback"
            )
        l = ctypes.CDLL(LIB_PATH)
        _bind(l, _SIGS)
        _bind(l, _OPTIONAL_SIGS)
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().mi355_last_error()
        raise NativeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    try:
        return _DT[d]
    except KeyError:
        raise NativeError(f"unsupported dtype {d}")


def ptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a tensor (None -> NULL).  Refuses pageable CPU tensors: the kernels only run
    on the GPU and there is deliberately no host implementation behind this ABI."""
    if t is None:
        return c_p(0)
    if not t.is_cuda and not t.is_pinned():
        raise NativeError("librecsys_amd expects GPU tensors (no CPU fallback exists)")
    # pinned host tensors are device-visible at the same address (hipHostMalloc, unified addressing): the host
    # tier of the embedding storage is read and written by the same kernels, over the host link
    return c_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def current_torch_stream():
    """torch.cuda.current_stream() of the current device WITHOUT the device-availability probe the default-argument form
    runs (torch.cuda.is_available -> hipGetDeviceCount, 5-10 us per call; the sharded step made ~7 such calls)."""
    if _cur_device is not None:
        return torch.cuda.current_stream(_cur_device())
    return torch.cuda.current_stream()


class on_stream:
    """`with torch.cuda.stream(s):` for a stream of the CURRENT device, without StreamContext's availability probes."""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        self.prev = current_torch_stream()
        torch.cuda.set_stream(self.s)
        return self.s

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


def stream() -> ctypes.c_void_p:
    """current HIP stream of the current device.  (torch.cuda.current_stream() costs ~6 us of Python per call -- ten native
    calls per step made that a visible part of the sharded step's host time; the raw-stream query is a plain C call.)"""
    if _raw_stream is not None and _cur_device is not None:
        return c_p(_raw_stream(_cur_device()))
    return c_p(torch.cuda.current_stream().cuda_stream)


def require_contiguous(*ts):
    for t in ts:
        if t is not None and not t.is_contiguous():
            raise NativeError("tensor must be contiguous")
