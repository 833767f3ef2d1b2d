"""`paged_kvcache_ops` of the reference (examples/commons/ops/cuda_ops/csrc/paged_kvcache_ops_cuda.cpp:325-337) on
MI355X: importing this module registers `torch.ops.paged_kvcache_ops.append_kvcache` with the reference's schema, so
`examples/hstu/modules/paged_hstu_infer_layer.py:350-364` calls it unchanged.  `gather_kvcache` (cache off-load) is part
of the KV-cache manager, which is out of scope."""
import torch

from hstu.hstu_attn_interface import append_kvcache  # noqa: F401

_lib = torch.library.Library("paged_kvcache_ops", "FRAGMENT")
_lib.define("append_kvcache(Tensor append_key, Tensor append_value, Tensor batch_indices, Tensor positions, "
            "Tensor seqlen_offsets, Tensor nnz_cuda, int max_nnz, Tensor(a!) kv_cache_table, Tensor kv_indices, "
            "Tensor kv_indptr, Tensor kv_last_page_len, int kv_layout) -> Tensor(a!)")


def _impl(append_key, append_value, batch_indices, positions, seqlen_offsets, nnz_cuda, max_nnz, kv_cache_table,
          kv_indices, kv_indptr, kv_last_page_len, kv_layout):
    return append_kvcache(append_key, append_value, batch_indices, positions, seqlen_offsets, nnz_cuda, max_nnz,
                          kv_cache_table, kv_indices, kv_indptr, kv_last_page_len, kv_layout)


_lib.impl("append_kvcache", _impl, "CUDA")
