"""Drop-in for the `hstu` package of the reference (pip `fbgemm_gpu_hstu`, built from the un-vendored
third_party/FBGEMM submodule): `hstu_attn_varlen_func` with the positional order the example pins
(examples/hstu/modules/hstu_attention.py:296-314, test/hstu_attn/test_hstu_attn_smoke.py:105-121), on top of
the gfx950 MFMA kernels (mi355_hstu_attn_fwd / mi355_hstu_attn_bwd).
"""
from .hstu_attn_interface import (HstuAttnVarlenFunc, HstuAttnWindowFunc, append_kvcache, hstu_attn_varlen_func,  # noqa: F401
                                  hstu_varlen_bwd, hstu_varlen_bwd_window, hstu_varlen_fwd, hstu_varlen_fwd_kv,
                                  hstu_varlen_fwd_window, HstuAttnRabFunc, hstu_varlen_fwd_rab, hstu_varlen_bwd_rab)

try:  # `import hstu` registers torch.ops.fbgemm.hstu_varlen_* (the example relies on it: fused_hstu_op.py:19)
    from . import hstu_ops_gpu  # noqa: F401
except Exception as _e:  # pragma: no cover - e.g. a torch build without torch.library
    import warnings

    warnings.warn(f"hstu: torch.ops.fbgemm.hstu_varlen_* were not registered ({_e})")
