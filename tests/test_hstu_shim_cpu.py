"""Host-side argument checks of the `hstu` drop-in (`hstu_attn_varlen_func`, recsys-examples_amd/hstu/hstu_attn_interface.py)
on CPU: the checks run before any device work, so the reference's error behaviour for the mask / bias options
(corelib/hstu/hstu_attn/hstu_attn_interface.py:234-255, hstu_api.cpp:154-165,417-430) is testable without a GPU.  The compute
itself is covered by tests/test_hstu_gpu.py through the C ABI."""
import pytest
import torch

from hstu import hstu_attn_varlen_func
from hstu.hstu_attn_interface import _check_inputs, _rab_strides


def _qkv(T=8, H=2, d=32, dtype=torch.bfloat16):
    q = torch.zeros(T, H, d, dtype=dtype)
    return q, q.clone(), q.clone(), torch.tensor([0, T], dtype=torch.int32)


def test_contexts_and_targets_need_the_causal_mask():
    q, k, v, cu = _qkv()
    one = torch.tensor([1], dtype=torch.int32)
    for window in ((-1, -1), (3, 0), (3, 2), (-1, 5)):
        with pytest.raises(ValueError, match="context is True and causal is not True"):
            hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 8, 8, 8, one, None, window_size=window)
        with pytest.raises(ValueError, match="target is True and causal is not True"):
            hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 8, 8, 8, None, one, window_size=window)
    # (-1, 0) with both is the ordinary training call: passes the checks
    assert _check_inputs(q, k, v, cu, cu, one, one, (-1, 0), None, None, None, None) is True


def test_window_normalisation_and_causal_flag():
    q, k, v, cu = _qkv()
    # negative sides are "unbounded" (hstu_api.cpp:154-155): (-5, 0) is the causal mask, (-1, -7) the full one
    assert _check_inputs(q, k, v, cu, cu, None, None, (-5, 0), None, None, None, None) is True
    assert _check_inputs(q, k, v, cu, cu, None, None, (-1, -7), None, None, None, None) is False
    # a finite side makes it a local window: not the plain causal path
    assert _check_inputs(q, k, v, cu, cu, None, None, (4, 0), None, None, None, None) is False


def test_dtype_shape_and_unsupported_options():
    q, k, v, cu = _qkv()
    with pytest.raises(RuntimeError, match="bf16"):
        hstu_attn_varlen_func(q.float(), k.float(), v.float(), cu, cu, None, None, 8, 8, 8, None, None)
    with pytest.raises(RuntimeError, match="head_dim"):
        q48 = torch.zeros(8, 2, 48, dtype=torch.bfloat16)
        hstu_attn_varlen_func(q48, q48, q48, cu, cu, None, None, 8, 8, 8, None, None)
    with pytest.raises(NotImplementedError, match="seqused"):
        hstu_attn_varlen_func(q, k, v, cu, cu, cu, None, 8, 8, 8, None, None)
    with pytest.raises(RuntimeError, match="int32"):
        hstu_attn_varlen_func(q, k, v, cu.long(), cu.long(), None, None, 8, 8, 8, None, None)
    with pytest.raises(RuntimeError, match="max_seqlen_q must be <= max_seqlen_k"):
        hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 9, 8, 8, None, None)


def test_rab_checks():
    q, k, v, cu = _qkv()
    with pytest.raises(ValueError, match="rab is None, but has_drab is True"):
        hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 8, 8, 8, None, None, has_drab=True)
    bad_heads = torch.zeros(1, 3, 8, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="Number of heads in rab must be 1 or equal"):
        hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 8, 8, 8, None, None, rab=bad_heads)
    with pytest.raises(RuntimeError, match="tensor of the dtype of q"):
        hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 8, 8, 8, None, None, rab=torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError, match="max_seqlen_k"):
        hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 8, 8, 8, None, None, rab=torch.zeros(1, 2, 9, 9, dtype=torch.bfloat16))
    # one shared bias head is a head stride of 0 for the kernels; per-head keeps its stride; slices keep their row stride
    shared = torch.zeros(3, 1, 8, 8, dtype=torch.bfloat16)
    per_head = torch.zeros(3, 2, 16, 16, dtype=torch.bfloat16)[:, :, :8, :8]
    assert _rab_strides(shared, 2) == (64, 0, 8)
    assert _rab_strides(per_head, 2) == (512, 256, 16)
