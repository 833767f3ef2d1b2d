"""autograd wrapper of the HSTU jagged attention kernels (MI355X).

Mirrors `HstuAttnVarlenFunc` / `hstu_attn_varlen_func` of the reference
(corelib/hstu/hstu_attn/hstu_attn_interface.py:23-279 legacy signature; new positional order of
examples/hstu/modules/hstu_attention.py:296-314): same argument meaning, same input checks
(bf16 or fp16 q / k / v, int32 cu_seqlens / num_contexts / num_targets, head_dim in {32, 64, 128, 256},
contextual / target masks require causal -- hstu_api.cpp:359-430).
Inference extensions (forward only): cu_seqlens_k longer than cu_seqlens_q (delta-q) and the paged KV cache
(kv_cache / page_offsets / page_ids / last_page_lens).
Arbitrary mask functions (`func`, hstu_api.cpp:170-180) are read INSIDE the kernels (mi355_hstu_attn_{fwd_kv,bwd}_func: no mask
tensor exists), forward and backward, with any other mask -- contextual rows keep their view of the history, as in the
reference's kernels (hstu_fwd.h:519-524) --, over delta-q / paged keys too.  Next to a relative bias they are added to it as a
0 / -1e9 bias (func_mask_bias: O(batch max_seqlen_k^2) memory, like the bias itself).
Not supported (raise): seqused_* (no caller of the reference passes them, and its kernels take none).  The raw ops of the fused layer
(`torch.ops.fbgemm.hstu_varlen_{fwd,bwd}_{80,90}`) are registered by `hstu.hstu_ops_gpu`.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch

import mi355_native as N
from mi355_native import c_f, c_i64, c_int, c_p, check, lib, ptr, stream

N.register_signatures({
    "mi355_hstu_attn_fwd": [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_i64,
                            c_i64, c_i64, c_p, c_p, c_i64, c_int, c_f, c_f, c_p],
    "mi355_hstu_attn_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p,
                            c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_int, c_f, c_f, c_p, c_i64, c_p],
    "mi355_hstu_attn_bwd_workspace_bytes": [c_i64, c_i64, c_i64],
    "mi355_hstu_attn_bwd_ds_bytes": [c_i64, c_i64, c_i64, c_i64],
    "mi355_hstu_attn_bwd_ds_bytes_capped": [c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int],
    "mi355_hstu_attn_bwd_hint_tokens": [c_i64],
    "mi355_hstu_attn_fwd_hint_tokens": [c_i64],
    "mi355_hstu_attn_fwd_rab": [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_i64,
                                c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_f, c_p, c_i64, c_i64, c_i64, c_p],
    "mi355_hstu_attn_bwd_rab": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p,
                                c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_f, c_p, c_i64, c_i64, c_i64,
                                c_p, c_i64, c_i64, c_i64, c_p],
    "mi355_hstu_attn_fwd_window": [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_i64,
                                   c_i64, c_i64, c_i64, c_i64, c_f, c_f, c_p],
    "mi355_hstu_attn_bwd_window": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p,
                                   c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_f, c_p, c_i64, c_p],
    "mi355_hstu_attn_fwd_kv": [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64,
                               c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_int, c_f, c_f, c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_hstu_attn_fwd_kv_window": [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64,
                                      c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_f, c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_hstu_attn_fwd_kv_rab": [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64,
                                   c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_f, c_p, c_i64, c_i64, c_i64,
                                   c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_hstu_attn_fwd_kv_func": [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64,
                                    c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_f, c_p, c_i64, c_i64, c_i64,
                                    c_f, c_p, c_p, c_p, c_p, c_i64, c_p],
    "mi355_hstu_attn_bwd_func": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p,
                                 c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_f, c_p, c_i64, c_i64, c_i64,
                                 c_f, c_p, c_i64, c_p, c_i64, c_p],
    "mi355_append_kvcache": [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p,
                             c_i64, c_i64, c_p],
}, {"mi355_hstu_attn_bwd_workspace_bytes": c_i64, "mi355_hstu_attn_bwd_ds_bytes": c_i64,
    "mi355_hstu_attn_bwd_ds_bytes_capped": c_i64, "mi355_hstu_attn_bwd_hint_tokens": None,
    "mi355_hstu_attn_fwd_hint_tokens": None, "mi355_hstu_attn_fwd_hint_tokens_f16": None})
# the fp16-operand twins of the seven type-specific entry points (same argument lists)
_TYPED = ("mi355_hstu_attn_fwd_hint_tokens", "mi355_hstu_attn_fwd", "mi355_hstu_attn_fwd_kv", "mi355_hstu_attn_fwd_kv_window", "mi355_hstu_attn_fwd_kv_rab", "mi355_hstu_attn_bwd", "mi355_hstu_attn_fwd_window",
          "mi355_hstu_attn_bwd_window", "mi355_hstu_attn_fwd_rab", "mi355_hstu_attn_bwd_rab", "mi355_hstu_attn_fwd_kv_func",
          "mi355_hstu_attn_bwd_func")
N.register_signatures({n + "_f16": N.signature_of(n) for n in _TYPED})


def _fn(name: str, t: torch.Tensor):
    """the entry point for the operand type of `t` (bf16: the plain name, fp16: its _f16 twin)"""
    return getattr(lib(), name + ("_f16" if t.dtype == torch.float16 else ""))



def _check_inputs(q, k, v, cu_q, cu_k, num_contexts, num_targets, window_size, rab, kv_cache, seqused_q, seqused_k):
    if rab is not None:
        # hstu_api.cpp:417-430: (batch, heads or 1, max_seqlen_k, max_seqlen_k), contiguous last dimension
        if rab.dtype != q.dtype or rab.dim() != 4 or rab.stride(-1) != 1:
            raise RuntimeError("rab must be a (batch, nheads or 1, max_seqlen_k, max_seqlen_k) tensor of the dtype of q with a contiguous last dimension")
        if rab.shape[0] != cu_q.numel() - 1 or rab.shape[1] not in (1, q.shape[1]) or rab.shape[2] != rab.shape[3]:
            raise RuntimeError("Number of heads in rab must be 1 or equal to number of heads in query; shape (batch, heads, max_seqlen_k, max_seqlen_k)")
    if seqused_q is not None or seqused_k is not None:
        raise NotImplementedError("seqused_q / seqused_k are not supported")
    if q.dtype not in (torch.bfloat16, torch.float16) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError("HSTU only supports fp16 and bf16 data type")      # (hstu_api.cpp:359-366)
    if q.dim() != 3 or k.dim() != 3 or k.shape[1:] != q.shape[1:] or v.shape != k.shape:
        raise RuntimeError("q, k, v must be (total, nheads, head_dim); k and v of equal shape")
    if q.stride(-1) != 1 or k.stride(-1) != 1 or v.stride(-1) != 1:
        raise RuntimeError("q, k, v must have a contiguous last dimension")
    if cu_q.dtype != torch.int32 or cu_k.dtype != torch.int32:
        raise RuntimeError("cu_seqlens must be int32")
    for t, name in ((num_contexts, "num_contexts"), (num_targets, "num_targets")):
        if t is not None and t.dtype != torch.int32:
            raise RuntimeError(f"{name} must be int32")
    wl, wr = window_size
    wl, wr = (-1 if wl < 0 else int(wl)), (-1 if wr < 0 else int(wr))
    # hstu_attn_interface.py:238-245 of the reference: contextual / target rows only with the plain causal mask
    if num_contexts is not None and (wl, wr) != (-1, 0):
        raise ValueError("AssertError: context is True and causal is not True, this is undefined behavior")
    if num_targets is not None and (wl, wr) != (-1, 0):
        raise ValueError("AssertError: target is True and causal is not True, this is undefined behavior")
    local = not (wl == -1 and wr in (-1, 0))
    causal = wr == 0 and not local
    if q.shape[-1] not in (32, 64, 128, 256):
        raise RuntimeError("head_dim must be one of 32, 64, 128, 256")
    return causal


def hstu_varlen_fwd(q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size,
                    causal, alpha):
    """Raw forward (stands in for torch.ops.fbgemm.hstu_varlen_fwd_80 / hstu_attn_2_cuda.varlen_fwd)."""
    T, H, D = q.shape
    out = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    B = cu_seqlens.numel() - 1
    _fn("mi355_hstu_attn_fwd_hint_tokens", q)(int(q.shape[0]))   # (dense batches take the paired-row-block kernel)
    check(_fn("mi355_hstu_attn_fwd", q)(ptr(q), ptr(k), ptr(v), ptr(out), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                    q.stride(1), k.stride(1), v.stride(1), out.stride(1), ptr(cu_seqlens), B, H, D,
                                    int(max_seqlen), ptr(num_contexts), ptr(num_targets), int(target_group_size), int(causal),
                                    c_f(alpha), c_f(float(scaling_seqlen)), stream()), "hstu_attn_fwd")
    return out


def hstu_varlen_fwd_kv(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, scaling_seqlen, num_contexts, num_targets,
                       target_group_size, causal, alpha, kv_cache=None, page_offsets=None, page_ids=None,
                       last_page_lens=None, window=None, rab=None, max_seqlen_k=None):
    """Inference forward: delta-q (cu_seqlens_k) and / or paged KV cache [num_pages, 2, page_size, H, d]; `window` = (left,
    right) of a local attention window over absolute positions (then no contextual / target rows); `rab`: relative attention
    bias [batch, heads or 1, max_seqlen_k, max_seqlen_k] over absolute positions (with `max_seqlen_k`)."""
    T, H, D = q.shape
    out = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    B = cu_seqlens_q.numel() - 1
    page_size = 0
    if kv_cache is not None:
        if kv_cache.dim() != 5 or kv_cache.size(1) != 2 or kv_cache.size(3) != H or kv_cache.size(4) != D:
            raise RuntimeError("kv_cache must be [num_pages, 2, page_size, nheads, head_dim]")
        if not kv_cache.is_contiguous() or kv_cache.dtype != q.dtype:
            raise RuntimeError("kv_cache must be contiguous and of the dtype of q")
        for t, name in ((page_offsets, "page_offsets"), (page_ids, "page_ids"), (last_page_lens, "last_page_lens")):
            if t is None or t.dtype != torch.int32:
                raise RuntimeError(f"{name} must be an int32 tensor")
        page_size = kv_cache.size(2)
    _fn("mi355_hstu_attn_fwd_hint_tokens", q)(int(q.shape[0]))   # (dense batches take the paired-row-block kernel)
    if rab is not None:
        wl_, wr_ = window if window is not None else (-1, 0 if causal else -1)
        check(_fn("mi355_hstu_attn_fwd_kv_rab", q)(ptr(q), ptr(k), ptr(v), ptr(out), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                                   q.stride(1), k.stride(1), v.stride(1), out.stride(1), ptr(cu_seqlens_q),
                                                   ptr(cu_seqlens_k), B, H, D, int(max_seqlen_q), int(max_seqlen_k),
                                                   ptr(num_contexts), ptr(num_targets), int(target_group_size), int(wl_), int(wr_),
                                                   c_f(alpha), c_f(float(scaling_seqlen)), ptr(rab), rab.stride(0),
                                                   rab.stride(1) if rab.shape[1] > 1 else 0, rab.stride(2), ptr(kv_cache),
                                                   ptr(page_offsets), ptr(page_ids), ptr(last_page_lens), page_size, stream()),
              "hstu_attn_fwd_kv_rab")
        return out
    if window is not None:
        check(_fn("mi355_hstu_attn_fwd_kv_window", q)(ptr(q), ptr(k), ptr(v), ptr(out), q.stride(0), k.stride(0), v.stride(0),
                                                      out.stride(0), q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                                      ptr(cu_seqlens_q), ptr(cu_seqlens_k), B, H, D, int(max_seqlen_q),
                                                      int(window[0]), int(window[1]), c_f(alpha), c_f(float(scaling_seqlen)),
                                                      ptr(kv_cache), ptr(page_offsets), ptr(page_ids), ptr(last_page_lens),
                                                      page_size, stream()), "hstu_attn_fwd_kv_window")
        return out
    check(_fn("mi355_hstu_attn_fwd_kv", q)(ptr(q), ptr(k), ptr(v), ptr(out), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                       q.stride(1), k.stride(1), v.stride(1), out.stride(1), ptr(cu_seqlens_q),
                                       ptr(cu_seqlens_k), B, H, D, int(max_seqlen_q), ptr(num_contexts), ptr(num_targets),
                                       int(target_group_size), int(causal), c_f(alpha), c_f(float(scaling_seqlen)),
                                       ptr(kv_cache), ptr(page_offsets), ptr(page_ids), ptr(last_page_lens), page_size,
                                       stream()), "hstu_attn_fwd_kv")
    return out


def append_kvcache(append_key, append_value, batch_indices, positions, seqlen_offsets, nnz_cuda, max_nnz, kv_cache_table,
                   kv_indices, kv_indptr, kv_last_page_len, kv_layout=0):
    """torch.ops.paged_kvcache_ops.append_kvcache (paged_kvcache_ops_cuda.cpp:332): writes the new-history tokens of
    append_key / append_value into the paged cache (NHD layout) and returns the table."""
    if kv_layout != 0:
        raise NotImplementedError("only the NHD layout is supported")
    H, D = append_key.shape[1], append_key.shape[2]
    check(lib().mi355_append_kvcache(ptr(kv_cache_table), ptr(kv_indices), ptr(kv_indptr), H, D, kv_cache_table.size(2),
                                     ptr(append_key), ptr(append_value), append_key.stride(0), append_value.stride(0),
                                     append_key.stride(1), append_value.stride(1), ptr(batch_indices), ptr(positions),
                                     ptr(seqlen_offsets), ptr(nnz_cuda), int(max_nnz), batch_indices.numel(), stream()),
          "append_kvcache")
    return kv_cache_table


# Byte cap of the backward's P / dS exchange.  Measured at d = 256, H = 4 (profiles/r04_hstu_exchange_cap.txt): 32 x 4096 backward
# 581 TFLOP/s at 1 GiB (5 chunks), 640 at 2 GiB, 691 at 4 GiB (2 chunks), 707 at 8 GiB and above (dense layout 8.6 GB: one pass);
# 8 x 4096 612 / 664 / 671 -- every chunk is three launches whose tails cannot be filled by the next chunk.  4 GiB = 1.4 % of the HBM.
_DS_MAX_BYTES = int(__import__("os").environ.get("MI355_HSTU_DS_MAX_BYTES", str(4 << 30)))


def _bwd_exchange_workspace(q, B, H, D, max_seqlen, plain_causal):
    """Scratch of the backward's P / dS exchange (the dK pass leaves them for the one-GEMM dV / dQ passes): the dense layout
    B x H x ceil(max_seqlen / 32)^2 tiles when that fits under MI355_HSTU_DS_MAX_BYTES (default 4 GiB), else the jagged,
    chunked layout of mi355_hstu_attn_bwd_ds_bytes_capped -- sized by the batch's own lengths, never above the cap, walked in
    chunks by the library.  No driver query, no host read of the lengths; the decision depends on the shapes only."""
    L = lib()
    dense = L.mi355_hstu_attn_bwd_ds_bytes(B, H, D, int(max_seqlen))
    n = dense if dense <= _DS_MAX_BYTES else L.mi355_hstu_attn_bwd_ds_bytes_capped(B, H, D, int(max_seqlen), int(q.shape[0]),
                                                                                 _DS_MAX_BYTES, int(bool(plain_causal)))
    ws = torch.empty(max(int(n), 256), dtype=torch.uint8, device=q.device)
    # the hint is consumed by the NEXT backward call of this thread: set it last, behind everything here that can raise (an
    # allocation failure used to leave it behind for a later backward of another batch -- round-4 advisor finding)
    L.mi355_hstu_attn_bwd_hint_tokens(int(q.shape[0]))
    return ws


def hstu_varlen_bwd(dout, q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size,
                    causal, alpha):
    """Raw backward (stands in for hstu_varlen_bwd_80 / varlen_bwd): returns (dq, dk, dv)."""
    T, H, D = q.shape
    dout = dout.contiguous() if dout.stride(-1) != 1 else dout
    dq = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    dk = torch.empty_like(dq)
    dv = torch.empty_like(dq)
    B = cu_seqlens.numel() - 1
    ws = _bwd_exchange_workspace(q, B, H, D, max_seqlen, bool(causal) and num_contexts is None)
    check(_fn("mi355_hstu_attn_bwd", q)(ptr(dout), ptr(q), ptr(k), ptr(v), ptr(dq), ptr(dk), ptr(dv), q.stride(0), k.stride(0),
                                    v.stride(0), dout.stride(0), q.stride(1), k.stride(1), v.stride(1), dout.stride(1),
                                    ptr(cu_seqlens), B, H, D, int(max_seqlen), ptr(num_contexts), ptr(num_targets),
                                    int(target_group_size), int(causal), c_f(alpha), c_f(float(scaling_seqlen)), ptr(ws),
                                    ws.numel(), stream()), "hstu_attn_bwd")
    return dq, dk, dv


class HstuAttnVarlenFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size, causal,
                alpha):
        out = hstu_varlen_fwd(q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size,
                              causal, alpha)
        ctx.save_for_backward(q, k, v, cu_seqlens, num_contexts, num_targets)
        ctx.meta = (max_seqlen, scaling_seqlen, target_group_size, causal, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, cu, nc, nt = ctx.saved_tensors
        max_seqlen, scaling, g, causal, alpha = ctx.meta
        dq, dk, dv = hstu_varlen_bwd(dout, q, k, v, cu, max_seqlen, scaling, nc, nt, g, causal, alpha)
        return dq, dk, dv, None, None, None, None, None, None, None, None


def hstu_varlen_fwd_window(q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, wl, wr, alpha):
    """Raw forward with a local window: query i sees keys i - wl .. i + wr (a negative side is unbounded)."""
    T, H, D = q.shape
    out = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    B = cu_seqlens.numel() - 1
    _fn("mi355_hstu_attn_fwd_hint_tokens", q)(int(q.shape[0]))   # (dense batches take the paired-row-block kernel)
    check(_fn("mi355_hstu_attn_fwd_window", q)(ptr(q), ptr(k), ptr(v), ptr(out), q.stride(0), k.stride(0), v.stride(0),
                                           out.stride(0), q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                           ptr(cu_seqlens), B, H, D, int(max_seqlen), int(wl), int(wr), c_f(alpha),
                                           c_f(float(scaling_seqlen)), stream()), "hstu_attn_fwd_window")
    return out


def hstu_varlen_bwd_window(dout, q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, wl, wr, alpha):
    """Raw backward with a local window: returns (dq, dk, dv); same optional dS / P exchange as hstu_varlen_bwd."""
    T, H, D = q.shape
    dout = dout.contiguous() if dout.stride(-1) != 1 else dout
    dq = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    dk, dv = torch.empty_like(dq), torch.empty_like(dq)
    B = cu_seqlens.numel() - 1
    ws = _bwd_exchange_workspace(q, B, H, D, max_seqlen, False)
    check(_fn("mi355_hstu_attn_bwd_window", q)(ptr(dout), ptr(q), ptr(k), ptr(v), ptr(dq), ptr(dk), ptr(dv), q.stride(0),
                                           k.stride(0), v.stride(0), dout.stride(0), q.stride(1), k.stride(1), v.stride(1),
                                           dout.stride(1), ptr(cu_seqlens), B, H, D, int(max_seqlen), int(wl), int(wr),
                                           c_f(alpha), c_f(float(scaling_seqlen)), ptr(ws), ws.numel(), stream()),
          "hstu_attn_bwd_window")
    return dq, dk, dv


class HstuAttnWindowFunc(torch.autograd.Function):
    """local (sliding window) attention: query i sees keys i - left .. i + right (mi355_hstu_attn_{fwd,bwd}_window)"""

    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, wl, wr, alpha):
        out = hstu_varlen_fwd_window(q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, wl, wr, alpha)
        ctx.save_for_backward(q, k, v, cu_seqlens)
        ctx.meta = (max_seqlen, scaling_seqlen, wl, wr, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, cu = ctx.saved_tensors
        max_seqlen, scaling, wl, wr, alpha = ctx.meta
        dq, dk, dv = hstu_varlen_bwd_window(dout, q, k, v, cu, max_seqlen, scaling, wl, wr, alpha)
        return dq, dk, dv, None, None, None, None, None, None


def _func_neg(dtype) -> float:
    """bias of a masked (query, key) pair: SiLU(alpha (q.k + bias)) is -0 exactly (the sigmoid underflows to 0; so does SiLU', so
    no gradient crosses the mask).  It has to stay FINITE in the operand type (-inf x 0 would be NaN): -1e9 in bf16 (any alpha
    above ~1e-7), -6e4 in fp16 (alpha above ~2e-3; 1 / sqrt(head_dim) is 0.06 .. 0.18)."""
    return -6.0e4 if dtype == torch.float16 else -1.0e9



def func_mask_bias(func, cu_seqlens_q, cu_seqlens_k, max_seqlen_k, dtype):
    """Arbitrary mask functions (`func`, hstu_api.cpp:170-180; applied in hstu_fwd.h:493-556) as an attention bias for the rab
    kernels: [batch, heads_func, N, N] (N = max_seqlen_k), 0 where query token t may see key column j -- j < func[h, 0, t], or
    func[h, 2p - 1, t] <= j < func[h, 2p, t] for a p >= 1 -- and _func_neg(dtype) elsewhere.  Row i of a sequence is the query token
    i - (Lk - Lq) (delta-q: the queries are the sequence's last rows, corelib/hstu/test.py:122-131).  Built on the device without
    a host read; O(batch N^2) memory, which is what buys the masks the whole machinery of the biased kernels (any other mask on
    top, gradients, delta-q / paged keys)."""
    if func.dtype != torch.int32 or func.dim() != 3 or func.shape[1] % 2 != 1 or func.stride(-1) != 1:
        raise RuntimeError("func must be an int32 (heads or 1, n_func, >= total_q) tensor with an odd n_func and a contiguous last dimension")
    dev = func.device
    N = int(max_seqlen_k)
    cq, ck = cu_seqlens_q.to(torch.int64), cu_seqlens_k.to(torch.int64)
    off = (ck[1:] - ck[:-1]) - (cq[1:] - cq[:-1])                               # [B] first query row of every sequence
    rows = torch.arange(N, device=dev)
    tok = (cq[:-1, None] + (rows[None, :] - off[:, None])).clamp_(0, func.shape[-1] - 1)   # [B, N] (rows outside a sequence: any token)
    f = func[:, :, tok].to(torch.int64)                                          # [Hf, n_func, B, N]
    j = rows.view(1, 1, 1, N)
    ok = j < f[:, 0, :, :, None]
    for p in range(1, func.shape[1] // 2 + 1):
        ok |= (f[:, 2 * p - 1, :, :, None] <= j) & (j < f[:, 2 * p, :, :, None])
    bias = torch.zeros(ok.shape, dtype=dtype, device=dev).masked_fill_(~ok, _func_neg(dtype))
    return bias.permute(1, 0, 2, 3).contiguous()                                 # [B, Hf, N, N]


def _rab_strides(rab, num_heads):
    """(batch, head, row) strides in elements; one shared bias head is a head stride of 0"""
    return rab.stride(0), (0 if rab.shape[1] == 1 else rab.stride(1)), rab.stride(2)


def hstu_varlen_fwd_rab(q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size, wl, wr,
                        alpha, rab):
    """Raw forward with a relative attention bias: SiLU(alpha (q.k + rab[b, h, i, j])) (hstu_api.cpp:417-430)."""
    T, H, D = q.shape
    out = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    B = cu_seqlens.numel() - 1
    rb, rh, rr = _rab_strides(rab, H)
    check(_fn("mi355_hstu_attn_fwd_rab", q)(ptr(q), ptr(k), ptr(v), ptr(out), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                        q.stride(1), k.stride(1), v.stride(1), out.stride(1), ptr(cu_seqlens), B, H, D,
                                        int(max_seqlen), ptr(num_contexts), ptr(num_targets), int(target_group_size), int(wl),
                                        int(wr), c_f(alpha), c_f(float(scaling_seqlen)), ptr(rab), rb, rh, rr, stream()),
          "hstu_attn_fwd_rab")
    return out


def hstu_varlen_bwd_rab(dout, q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size,
                        wl, wr, alpha, rab, has_drab):
    """Raw backward with a relative attention bias: (dq, dk, dv, drab or None).  drab has the shape of rab
    (hstu_api.cpp:659-667, zero outside the sequences / the mask); with one shared bias head it is the sum over the heads,
    formed in fp32 from per-head matrices (the reference adds bf16 pairs atomically: same value, no fixed order)."""
    T, H, D = q.shape
    dout = dout.contiguous() if dout.stride(-1) != 1 else dout
    dq = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    dk, dv = torch.empty_like(dq), torch.empty_like(dq)
    B = cu_seqlens.numel() - 1
    N = rab.shape[-1]
    rb, rh, rr = _rab_strides(rab, H)
    drab = torch.zeros((B, H, N, N), dtype=q.dtype, device=q.device) if has_drab else None
    ds = (drab.stride(0), drab.stride(1), drab.stride(2)) if has_drab else (0, 0, 0)
    check(_fn("mi355_hstu_attn_bwd_rab", q)(ptr(dout), ptr(q), ptr(k), ptr(v), ptr(dq), ptr(dk), ptr(dv), q.stride(0), k.stride(0),
                                        v.stride(0), dout.stride(0), q.stride(1), k.stride(1), v.stride(1), dout.stride(1),
                                        ptr(cu_seqlens), B, H, D, int(max_seqlen), ptr(num_contexts), ptr(num_targets),
                                        int(target_group_size), int(wl), int(wr), c_f(alpha), c_f(float(scaling_seqlen)),
                                        ptr(rab), rb, rh, rr, ptr(drab), ds[0], ds[1], ds[2], stream()), "hstu_attn_bwd_rab")
    if has_drab and rab.shape[1] == 1 and H > 1:
        drab = drab.float().sum(1, keepdim=True).to(q.dtype)
    return dq, dk, dv, drab


_FUNC_NEG = {}
_FUNC_DENSE = __import__("os").environ.get("MI355_HSTU_FUNC_DENSE", "0") == "1"


def _func_neg_value(dtype) -> float:
    """_func_neg(dtype) as the operand type holds it (what the dense-bias statement of the mask adds)"""
    v = _FUNC_NEG.get(dtype)
    if v is None:
        v = _FUNC_NEG[dtype] = float(torch.tensor(_func_neg(dtype), dtype=dtype))
    return v


def _check_func(func, q):
    if func.dtype != torch.int32 or func.dim() != 3 or func.shape[1] % 2 != 1 or func.stride(-1) != 1:
        raise RuntimeError("func must be an int32 (heads or 1, n_func, >= total_q) tensor with an odd n_func and a contiguous last dimension")
    if func.shape[0] not in (1, q.shape[1]) or func.shape[-1] < q.shape[0]:
        raise RuntimeError("func must be (heads or 1, n_func, >= total_q)")


def hstu_varlen_fwd_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, scaling_seqlen, num_contexts, num_targets,
                         target_group_size, wl, wr, alpha, func, kv_cache=None, page_offsets=None, page_ids=None, last_page_lens=None):
    """Raw forward with arbitrary mask functions read inside the kernel (mi355_hstu_attn_fwd_kv_func): no dense bias.
    cu_seqlens_k None = the keys are the queries' tokens (training)."""
    T, H, D = q.shape
    out = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    B = cu_seqlens_q.numel() - 1
    page_size = kv_cache.size(2) if kv_cache is not None else 0
    _fn("mi355_hstu_attn_fwd_hint_tokens", q)(int(q.shape[0]))   # (dense batches take the paired-row-block kernel)
    check(_fn("mi355_hstu_attn_fwd_kv_func", q)(ptr(q), ptr(k), ptr(v), ptr(out), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                                q.stride(1), k.stride(1), v.stride(1), out.stride(1), ptr(cu_seqlens_q),
                                                ptr(cu_seqlens_k), B, H, D, int(max_seqlen_q), int(max_seqlen_k),
                                                ptr(num_contexts), ptr(num_targets), int(target_group_size), int(wl), int(wr),
                                                c_f(alpha), c_f(float(scaling_seqlen)), ptr(func),
                                                func.stride(0) if func.shape[0] > 1 else 0, func.stride(1), func.shape[1],
                                                c_f(_func_neg_value(q.dtype)), ptr(kv_cache), ptr(page_offsets), ptr(page_ids),
                                                ptr(last_page_lens), page_size, stream()), "hstu_attn_fwd_kv_func")
    return out


def hstu_varlen_bwd_func(dout, q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size, wl, wr,
                         alpha, func):
    """Raw backward with arbitrary mask functions read inside the kernels: (dq, dk, dv)"""
    T, H, D = q.shape
    dout = dout.contiguous() if dout.stride(-1) != 1 else dout
    dq = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    dk, dv = torch.empty_like(dq), torch.empty_like(dq)
    B = cu_seqlens.numel() - 1
    # tile skipping (round 6): room for the key-block table the backward fills in front of its passes -- which query rows reach
    # every 128-key block, derived from the extents of the functions (72 bytes per key block and function set: the block's query range + the extents of its four 32-row groups; MI355_HSTU_WSKIP=0, the
    # A/B switch of every tile-range clipping, turns it off in the library)
    fws = torch.empty((func.shape[0] if func.shape[0] > 1 else 1) * (T // 128 + B + 1) * 18, dtype=torch.int32, device=q.device)
    # functions of up to two bands take the P / dS exchange backward at head dim 256 (the scratch of hstu_varlen_bwd)
    ws = _bwd_exchange_workspace(q, B, H, D, max_seqlen, False) if (D == 256 and func.shape[1] <= 5) else None
    check(_fn("mi355_hstu_attn_bwd_func", q)(ptr(dout), ptr(q), ptr(k), ptr(v), ptr(dq), ptr(dk), ptr(dv), q.stride(0), k.stride(0),
                                             v.stride(0), dout.stride(0), q.stride(1), k.stride(1), v.stride(1), dout.stride(1),
                                             ptr(cu_seqlens), B, H, D, int(max_seqlen), ptr(num_contexts), ptr(num_targets),
                                             int(target_group_size), int(wl), int(wr), c_f(alpha), c_f(float(scaling_seqlen)),
                                             ptr(func), func.stride(0) if func.shape[0] > 1 else 0, func.stride(1), func.shape[1],
                                             c_f(_func_neg_value(q.dtype)), ptr(fws), fws.numel() * 4 if fws is not None else 0,
                                             ptr(ws), ws.numel() if ws is not None else 0, stream()), "hstu_attn_bwd_func")
    return dq, dk, dv


class HstuAttnFuncFunc(torch.autograd.Function):
    """attention under arbitrary mask functions, the functions read inside the kernels (hstu_fwd.h:139-145, 493-556)"""

    @staticmethod
    def forward(ctx, q, k, v, func, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size, wl, wr,
                alpha):
        out = hstu_varlen_fwd_func(q, k, v, cu_seqlens, None, max_seqlen, max_seqlen, scaling_seqlen, num_contexts, num_targets,
                                   target_group_size, wl, wr, alpha, func)
        ctx.save_for_backward(q, k, v, func, cu_seqlens, num_contexts, num_targets)
        ctx.meta = (max_seqlen, scaling_seqlen, target_group_size, wl, wr, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, func, cu, nc, nt = ctx.saved_tensors
        max_seqlen, scaling, g, wl, wr, alpha = ctx.meta
        dq, dk, dv = hstu_varlen_bwd_func(dout, q, k, v, cu, max_seqlen, scaling, nc, nt, g, wl, wr, alpha, func)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None


class HstuAttnRabFunc(torch.autograd.Function):
    """attention with a relative bias; rab receives a gradient when has_drab (hstu_attn_interface.py:23-183 of the reference)"""

    @staticmethod
    def forward(ctx, q, k, v, rab, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets, target_group_size, wl,
                wr, alpha, has_drab):
        out = hstu_varlen_fwd_rab(q, k, v, cu_seqlens, max_seqlen, scaling_seqlen, num_contexts, num_targets,
                                  target_group_size, wl, wr, alpha, rab)
        ctx.save_for_backward(q, k, v, rab, cu_seqlens, num_contexts, num_targets)
        ctx.meta = (max_seqlen, scaling_seqlen, target_group_size, wl, wr, alpha, has_drab)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, rab, cu, nc, nt = ctx.saved_tensors
        max_seqlen, scaling, g, wl, wr, alpha, has_drab = ctx.meta
        dq, dk, dv, drab = hstu_varlen_bwd_rab(dout, q, k, v, cu, max_seqlen, scaling, nc, nt, g, wl, wr, alpha, rab, has_drab)
        return dq, dk, dv, drab, None, None, None, None, None, None, None, None, None, None


def hstu_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k, max_seqlen_q, max_seqlen_k,
                          scaling_seqlen, num_contexts, num_targets, target_group_size=1, window_size=(-1, -1), alpha=1.0,
                          rab=None, has_drab=False, kv_cache=None, page_offsets=None, page_ids=None, last_page_lens=None,
                          func=None, quant_mode=-1):
    """out (total_q, nheads, head_dim) = HSTU attention over the jagged batch described by cu_seqlens."""
    causal = _check_inputs(q, k, v, cu_seqlens_q, cu_seqlens_k, num_contexts, num_targets, window_size, rab, kv_cache,
                           seqused_q, seqused_k)
    if max_seqlen_q > max_seqlen_k:
        raise RuntimeError("max_seqlen_q must be <= max_seqlen_k")
    if scaling_seqlen is None or scaling_seqlen == -1:
        scaling_seqlen = max_seqlen_q
    # Self-attention (training) or delta-q / paged KV (inference)?  Decided from what the HOST knows -- the example passes
    # two distinct `offsets.to(int32)` tensors for the same offsets (hstu_attention.py:299-300), and reading them back to
    # compare would be a blocking device-to-host copy per attention layer per step (and impossible under graph capture).
    # Equal shapes of q / k and of the offset arrays with equal max lengths is self-attention; a delta-q call has more
    # keys than queries.
    same = kv_cache is None and (cu_seqlens_q.data_ptr() == cu_seqlens_k.data_ptr() or (
        cu_seqlens_q.shape == cu_seqlens_k.shape and q.shape[0] == k.shape[0] and int(max_seqlen_q) == int(max_seqlen_k)))
    wl, wr = (-1 if window_size[0] < 0 else int(window_size[0])), (-1 if window_size[1] < 0 else int(window_size[1]))
    if has_drab and rab is None:   # hstu_attn_interface.py:234-237 of the reference
        raise ValueError("AssertError: rab is None, but has_drab is True, is not allowed in backward")
    if func is not None and rab is None:
        # arbitrary mask functions, read inside the kernels (round 5; MI355_HSTU_FUNC_DENSE=1 keeps the dense-bias statement below,
        # the two agree bit for bit): they narrow whatever other mask applies, as in the reference (hstu_fwd.h:519-556) -- except
        # for the history columns of contextual rows, which its context test exempts
        _check_func(func, q)
        if not _FUNC_DENSE:
            if not same:   # inference (delta-q keys and / or the paged cache): forward only
                if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
                    raise NotImplementedError("delta-q / paged-KV attention is forward only (as in the reference's inference path)")
                return hstu_varlen_fwd_func(q, k, v, cu_seqlens_q, cu_seqlens_k, int(max_seqlen_q), int(max_seqlen_k), scaling_seqlen,
                                            num_contexts, num_targets, int(target_group_size), wl, wr, float(alpha), func, kv_cache,
                                            page_offsets, page_ids, last_page_lens)
            return HstuAttnFuncFunc.apply(q, k, v, func, cu_seqlens_q, int(max_seqlen_k), scaling_seqlen, num_contexts, num_targets,
                                          int(target_group_size), wl, wr, float(alpha))
    if func is not None:
        # with a relative bias as well (or MI355_HSTU_FUNC_DENSE=1): a bias of 0 / -1e9 through the biased kernels (func_mask_bias)
        if num_contexts is not None:
            raise NotImplementedError("func together with num_contexts needs the in-kernel mask functions (no rab)")
        _check_func(func, q)
        fb = func_mask_bias(func, cu_seqlens_q, cu_seqlens_k, max_seqlen_k, q.dtype)
        if rab is not None and rab.shape[-1] != int(max_seqlen_k):
            raise RuntimeError("rab must be (batch, nheads or 1, max_seqlen_k, max_seqlen_k)")
        rab = fb if rab is None else (rab + fb).clamp_(min=torch.finfo(q.dtype).min)   # (fp16: the sum must stay finite)
    if rab is not None:
        if rab.shape[-1] != int(max_seqlen_k):
            raise RuntimeError("rab must be (batch, nheads or 1, max_seqlen_k, max_seqlen_k)")
        if not same:   # inference (delta-q keys and / or the paged cache) with a bias: forward only
            if has_drab or (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad or rab.requires_grad)):
                raise NotImplementedError("delta-q / paged-KV attention is forward only (as in the reference's inference path)")
            local = not (wl == -1 and wr in (-1, 0))
            return hstu_varlen_fwd_kv(q, k, v, cu_seqlens_q, cu_seqlens_k, int(max_seqlen_q), scaling_seqlen, num_contexts, num_targets,
                                      int(target_group_size), causal, float(alpha), kv_cache, page_offsets, page_ids, last_page_lens,
                                      window=(wl, wr) if local else None, rab=rab, max_seqlen_k=int(max_seqlen_k))
        if rab.shape[-1] != int(max_seqlen_k):
            raise RuntimeError("rab must be (batch, nheads or 1, max_seqlen_k, max_seqlen_k)")
        return HstuAttnRabFunc.apply(q, k, v, rab, cu_seqlens_q, int(max_seqlen_k), scaling_seqlen, num_contexts, num_targets,
                                     int(target_group_size), wl, wr, float(alpha), bool(has_drab))
    if not (wl == -1 and wr in (-1, 0)):
        if not same:
            # inference under a local window (delta-q keys and / or the paged cache): forward only, as the plain inference path
            if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
                raise NotImplementedError("delta-q / paged-KV attention is forward only (as in the reference's inference path)")
            return hstu_varlen_fwd_kv(q, k, v, cu_seqlens_q, cu_seqlens_k, int(max_seqlen_q), scaling_seqlen, None, None, 1,
                                      wr == 0, float(alpha), kv_cache, page_offsets, page_ids, last_page_lens, window=(wl, wr))
        return HstuAttnWindowFunc.apply(q, k, v, cu_seqlens_q, int(max_seqlen_k), scaling_seqlen, wl, wr, float(alpha))
    if kv_cache is not None or not same:
        # inference: keys longer than the queries and / or history keys in the paged cache; no backward
        if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
            raise NotImplementedError("delta-q / paged-KV attention is forward only (as in the reference's inference path)")
        return hstu_varlen_fwd_kv(q, k, v, cu_seqlens_q, cu_seqlens_k, int(max_seqlen_q), scaling_seqlen, num_contexts,
                                  num_targets, int(target_group_size), causal, float(alpha), kv_cache, page_offsets,
                                  page_ids, last_page_lens)
    return HstuAttnVarlenFunc.apply(q, k, v, cu_seqlens_q, int(max_seqlen_k), scaling_seqlen, num_contexts, num_targets,
                                    int(target_group_size), causal, float(alpha))
