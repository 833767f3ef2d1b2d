"""`torch.ops.fbgemm.hstu_varlen_{fwd,bwd}_{80,90}` on MI355X.

The reference's fused HSTU layer does not go through `hstu_attn_varlen_func`: it calls the raw ops of the `hstu` pip package
(examples/hstu/ops/fused_hstu_op.py:318-366 forward, :682-750 backward), picking the `_80` or `_90` flavour from
`torch.cuda.get_device_properties(0).major` -- which is 9 on gfx950, so an unchanged example lands on the `_90` names here.
Importing this module (the example does: `import hstu.hstu_ops_gpu`, fused_hstu_op.py:19-20) defines all four ops with the
positional order of those call sites, backed by the gfx950 kernels, plus Meta kernels for export.  Arguments this build has
no kernel for (seqused, fp8 quantisation) must be None / default; the ops raise otherwise.  `func` (arbitrary mask functions)
is read inside the kernels (mi355_hstu_attn_{fwd_kv,bwd}_func); next to a relative bias it is added to it as a 0 / -1e9 bias
(hstu_attn_interface.func_mask_bias).
`window_size_left / right` with a finite side run the local-window kernels, `rab` / `has_drab` the bias kernels."""
import torch

from .hstu_attn_interface import (_check_func, func_mask_bias, hstu_varlen_bwd_func, hstu_varlen_fwd_func, hstu_varlen_bwd, hstu_varlen_bwd_rab, hstu_varlen_bwd_window, hstu_varlen_fwd,
                                  hstu_varlen_fwd_rab, hstu_varlen_fwd_window)

_T = "Tensor"
_O = "Tensor?"


def _already_there(name: str) -> bool:
    try:
        getattr(torch.ops.fbgemm, name)
        return True
    except (AttributeError, RuntimeError):
        return False


def _check(q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, wl, wr, rab, func, quant_mode=-1, extra=(),
           num_contexts=None, num_targets=None):
    if seqused_q is not None or seqused_k is not None:
        raise NotImplementedError("seqused_q / seqused_k are not supported")
    if func is not None and num_contexts is not None and rab is not None:
        raise NotImplementedError("func together with num_contexts needs the in-kernel mask functions (no rab)")
    if rab is not None and (rab.dim() != 4 or rab.shape[1] not in (1, q.shape[1]) or rab.shape[-1] != max_k or rab.stride(-1) != 1):
        raise RuntimeError("rab must be (batch, nheads or 1, max_seqlen_k, max_seqlen_k) with a contiguous last dimension")
    if quant_mode not in (-1, None) or any(e is not None for e in extra):
        raise NotImplementedError("fp8 quantisation is not supported")
    if q.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("HSTU only supports fp16 and bf16 data type")
    if max_q != max_k or q.shape[0] != k.shape[0] or cu_q.shape != cu_k.shape:
        raise NotImplementedError("the raw training ops are self-attention only (cu_seqlens_q == cu_seqlens_k)")
    if v.shape != k.shape:
        raise RuntimeError("v must have the shape of k (the fused layer asserts linear_dim == attention_dim, hstu_attention.py:248-250)")
    wl, wr = (-1 if wl < 0 else int(wl)), (-1 if wr < 0 else int(wr))
    window = None if (wl == -1 and wr in (-1, 0)) else (wl, wr)
    if (num_contexts is not None or num_targets is not None) and (wl, wr) != (-1, 0):
        # hstu_api.cpp:163-164
        raise ValueError("context / target masks need the causal mask (-1, 0): undefined behaviour otherwise")
    return wr == 0 and window is None, window


def _fwd(q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, scaling_seqlen, num_contexts, num_targets, target_group_size,
         wl, wr, alpha, rab, func, quant_mode=-1, output_dtype=0):
    causal, window = _check(q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, wl, wr, rab, func, quant_mode,
                            num_contexts=num_contexts, num_targets=num_targets)
    if output_dtype not in (0, None):
        raise NotImplementedError("output_dtype must be 0 (bf16)")
    if func is not None and rab is None:   # arbitrary mask functions, read inside the kernel
        _check_func(func, q)
        return hstu_varlen_fwd_func(q, k, v, cu_q, None, int(max_q), int(max_k), scaling_seqlen, num_contexts, num_targets,
                                    int(target_group_size), max(int(wl), -1), max(int(wr), -1), float(alpha), func), None
    if func is not None:   # ... next to a relative bias: a 0 / -1e9 bias added to it (hstu_attn_interface.func_mask_bias)
        fb = func_mask_bias(func, cu_q, cu_k, int(max_k), q.dtype)
        out = hstu_varlen_fwd_rab(q, k, v, cu_q, int(max_k), scaling_seqlen, num_contexts, num_targets, int(target_group_size),
                                  max(int(wl), -1), max(int(wr), -1), float(alpha), fb if rab is None else (rab + fb).clamp_(min=torch.finfo(q.dtype).min))
        return out, rab
    if rab is not None:
        return hstu_varlen_fwd_rab(q, k, v, cu_q, int(max_k), scaling_seqlen, num_contexts, num_targets, int(target_group_size),
                                   max(int(wl), -1), max(int(wr), -1), float(alpha), rab), rab
    if window is not None:
        return hstu_varlen_fwd_window(q, k, v, cu_q, int(max_k), scaling_seqlen, window[0], window[1], float(alpha)), None
    out = hstu_varlen_fwd(q, k, v, cu_q, int(max_k), scaling_seqlen, num_contexts, num_targets, int(target_group_size), causal,
                          float(alpha))
    return out, None


def _bwd(dout, q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, scaling_seqlen, dq, dk, dv, num_contexts, num_targets,
         target_group_size, wl, wr, alpha, rab, has_drab, func, deterministic, quant_mode=-1, extra=()):
    causal, window = _check(q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, wl, wr, rab, func, quant_mode, extra,
                            num_contexts=num_contexts, num_targets=num_targets)
    if has_drab and rab is None:
        raise RuntimeError("rab must exist when using has_drab")   # hstu_api.cpp:660
    drab = None
    if func is not None and rab is None:
        _check_func(func, q)
        g = hstu_varlen_bwd_func(dout, q, k, v, cu_q, int(max_k), scaling_seqlen, num_contexts, num_targets, int(target_group_size),
                                 max(int(wl), -1), max(int(wr), -1), float(alpha), func)
    elif func is not None:
        fb = func_mask_bias(func, cu_q, cu_k, int(max_k), q.dtype)
        *g, drab = hstu_varlen_bwd_rab(dout, q, k, v, cu_q, int(max_k), scaling_seqlen, num_contexts, num_targets,
                                       int(target_group_size), max(int(wl), -1), max(int(wr), -1), float(alpha),
                                       fb if rab is None else (rab + fb).clamp_(min=torch.finfo(q.dtype).min), bool(has_drab))
        if drab is not None and rab.shape[1] == 1 and drab.shape[1] > 1:   # (a per-head mask over one shared bias head)
            drab = drab.float().sum(1, keepdim=True).to(q.dtype)
    elif rab is not None:
        *g, drab = hstu_varlen_bwd_rab(dout, q, k, v, cu_q, int(max_k), scaling_seqlen, num_contexts, num_targets,
                                       int(target_group_size), max(int(wl), -1), max(int(wr), -1), float(alpha), rab,
                                       bool(has_drab))
    elif window is not None:
        g = hstu_varlen_bwd_window(dout, q, k, v, cu_q, int(max_k), scaling_seqlen, window[0], window[1], float(alpha))
    else:
        g = hstu_varlen_bwd(dout, q, k, v, cu_q, int(max_k), scaling_seqlen, num_contexts, num_targets,
                            int(target_group_size), causal, float(alpha))   # (deterministic by construction: no atomics)
    res = []
    for given, new in zip((dq, dk, dv), g):
        if given is not None:
            given.copy_(new)
            new = given
        res.append(new)
    return res[0], res[1], res[2], drab


def _fwd_90(q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, scaling_seqlen, num_contexts, num_targets,
            target_group_size, wl, wr, alpha, rab, func, quant_mode, output_dtype):
    return _fwd(q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, scaling_seqlen, num_contexts, num_targets,
                target_group_size, wl, wr, alpha, rab, func, quant_mode, output_dtype)


def _bwd_90(dout, dout_t, q, q_t, k, k_t, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, scaling_seqlen, dq, dk, dv,
            num_contexts, num_targets, target_group_size, wl, wr, alpha, quant_mode, rab, has_drab, func, d0, d1, d2, d3, d4,
            d5, d6, d7, d8, d9, d10, output_dtype, deterministic):
    return _bwd(dout, q, k, v, cu_q, cu_k, seqused_q, seqused_k, max_q, max_k, scaling_seqlen, dq, dk, dv, num_contexts,
                num_targets, target_group_size, wl, wr, alpha, rab, has_drab, func, deterministic, quant_mode,
                (dout_t, q_t, k_t, d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10))


def _fwd_meta(q, *a, **kw):
    return torch.empty_like(q), None


def _bwd80_meta(dout, q, k, v, *a, **kw):
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), None


def _bwd90_meta(dout, dout_t, q, q_t, k, k_t, v, *a, **kw):
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), None


_FWD_ARGS = (f"{_T} q, {_T} k, {_T} v, {_T} cu_seqlens_q, {_T} cu_seqlens_k, {_O} seqused_q, {_O} seqused_k, int max_seqlen_q, "
             f"int max_seqlen_k, float scaling_seqlen, {_O} num_contexts, {_O} num_targets, int target_group_size, "
             f"int window_size_left, int window_size_right, float alpha, {_O} rab, {_O} func")
_BWD_TAIL = (f"{_T} cu_seqlens_q, {_T} cu_seqlens_k, {_O} seqused_q, {_O} seqused_k, int max_seqlen_q, int max_seqlen_k, "
             f"float scaling_seqlen, Tensor(a!)? dq, Tensor(b!)? dk, Tensor(c!)? dv, {_O} num_contexts, {_O} num_targets, "
             "int target_group_size, int window_size_left, int window_size_right, float alpha")
_DESCALE = ", ".join(f"{_O} descale_{i}" for i in range(11))

_lib = None
if not _already_there("hstu_varlen_fwd_80"):
    _lib = torch.library.Library("fbgemm", "FRAGMENT")
    _lib.define(f"hstu_varlen_fwd_80({_FWD_ARGS}) -> (Tensor, Tensor?)")
    _lib.define(f"hstu_varlen_fwd_90({_FWD_ARGS}, int quant_mode, int output_dtype) -> (Tensor, Tensor?)")
    _lib.define(f"hstu_varlen_bwd_80({_T} dout, {_T} q, {_T} k, {_T} v, {_BWD_TAIL}, {_O} rab, bool has_drab, {_O} func, "
                "bool deterministic) -> (Tensor, Tensor, Tensor, Tensor?)")
    _lib.define(f"hstu_varlen_bwd_90({_T} dout, {_O} dout_t, {_T} q, {_O} q_t, {_T} k, {_O} k_t, {_T} v, {_BWD_TAIL}, "
                f"int quant_mode, {_O} rab, bool has_drab, {_O} func, {_DESCALE}, int output_dtype, bool deterministic) "
                "-> (Tensor, Tensor, Tensor, Tensor?)")
    for name, impl, meta in (("hstu_varlen_fwd_80", _fwd, _fwd_meta), ("hstu_varlen_fwd_90", _fwd_90, _fwd_meta),
                             ("hstu_varlen_bwd_80", _bwd, _bwd80_meta), ("hstu_varlen_bwd_90", _bwd_90, _bwd90_meta)):
        _lib.impl(name, impl, "CUDA")
        _lib.impl(name, meta, "Meta")
