"""GPU parity tests of the HSTU jagged attention kernels (forward + backward) through the `hstu` drop-in
package.  Acceptance rule = the reference's own (examples/commons/utils/hstu_assert_close.py:20-56):
max|out - ref_fp32| <= 2 x max|ref_bf16 - ref_fp32| forward, 5 x backward, against (a) golden vectors made
by the reference's pytorch_hstu_mha and (b) the CPU oracle on larger random jagged batches; plus
BASELINE.json's 1e-3 relative bound on the outputs (relative to the output scale)."""
import os

import numpy as np
import pytest
import torch

from oracle import hstu_oracle as ho

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "hstu_golden.npz"))
CASES = [str(c) for c in G["cases"]]


def _bf(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV).to(torch.bfloat16)


def _run(q, k, v, off, N, targets, ctx, grp, causal, alpha, dout=None, scaling=None):
    from hstu import hstu_attn_varlen_func

    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    cu = torch.from_numpy(np.asarray(off, np.int32)).to(DEV)
    nt = None if targets is None else torch.from_numpy(np.asarray(targets, np.int32)).to(DEV)
    nc = None if ctx is None else torch.from_numpy(np.asarray(ctx, np.int32)).to(DEV)
    out = hstu_attn_varlen_func(qq, kk, vv, cu, cu, None, None, N, N, scaling if scaling is not None else N, nc, nt,
                                target_group_size=grp, window_size=(-1, 0) if causal else (-1, -1), alpha=alpha)
    if dout is None:
        return out, None
    out.backward(dout)
    return out, (qq.grad, kk.grad, vv.grad)


def _close(actual, ref16, ref32, mult):
    a = actual.detach().float().cpu().numpy().reshape(-1)
    left = np.abs(a - ref32.reshape(-1)).max()
    right = np.abs(ref16.reshape(-1) - ref32.reshape(-1)).max()
    assert left <= mult * right + 1e-6, f"max|actual-ref32|={left:.3e} > {mult} x {right:.3e}"


@pytest.mark.parametrize("name", CASES)
def test_golden_fwd_bwd(name):
    g = lambda k: G[f"{name}/{k}"]
    H, d, causal, grp, N = [int(x) for x in g("meta")]
    t, c = g("targets"), g("ctx")
    targets = None if t[0] < 0 else t
    ctx = None if c[0] < 0 else c
    out, grads = _run(_bf(g("q")), _bf(g("k")), _bf(g("v")), g("off"), N, targets, ctx, grp, bool(causal), 1.0 / d ** 0.5,
                      dout=_bf(g("dout")))
    _close(out, g("out_bf16"), g("out"), 2)
    # BASELINE.json: within 1e-3 relative on bf16 HSTU outputs (relative to the output scale; bf16 ulp is 3.9e-3)
    scale = np.abs(g("out")).max()
    assert np.abs(out.detach().float().cpu().numpy() - g("out")).max() <= 4e-3 * scale
    _close(grads[0], g("dq_bf16"), g("dq"), 5)
    _close(grads[1], g("dk_bf16"), g("dk"), 5)
    _close(grads[2], g("dv_bf16"), g("dv"), 5)


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("mode", ["causal", "ctx_targets", "noncausal"])
def test_random_jagged_vs_oracle(d, mode):
    rng = np.random.default_rng(d + len(mode))
    B, H, maxL = 6, 2, 300
    lengths = rng.integers(1, maxL + 1, size=B)
    lengths[0] = maxL
    lengths[1] = 1
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    mk = lambda lo, hi: torch.empty(T, H, d, device=DEV).uniform_(lo, hi).bfloat16()
    q, k, v, dout = mk(-1, 1), mk(-1, 1), mk(-1, 1), mk(0, 1)
    targets = ctx = None
    if mode == "ctx_targets":
        targets = np.minimum(rng.integers(0, 11, size=B), lengths - 1)
        ctx = np.minimum(rng.integers(0, 5, size=B), np.maximum(lengths - 1 - targets, 0))
    causal = mode != "noncausal"
    alpha = 1.0 / d ** 0.5
    out, grads = _run(q, k, v, off, maxL, targets, ctx, 2 if mode == "ctx_targets" else 1, causal, alpha, dout=dout)
    qn, kn, vn, dn = (t.float().cpu().numpy() for t in (q, k, v, dout))
    grp = 2 if mode == "ctx_targets" else 1
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, maxL, causal, targets, ctx, grp)
    dq, dk, dv = ho.hstu_attn_bwd(dn, qn, kn, vn, off, alpha, maxL, causal, targets, ctx, grp)
    for got, want, tol in ((out, ref, 6e-3), (grads[0], dq, 1.2e-2), (grads[1], dk, 1.2e-2), (grads[2], dv, 1.2e-2)):
        gn = got.detach().float().cpu().numpy()
        err = np.abs(gn - want).max()
        assert err <= tol * np.abs(want).max() + 1e-6, f"{err} vs scale {np.abs(want).max()}"


def test_strided_inputs_and_scaling_seqlen():
    """q/k/v as slices of one fused [T, 3, H, d] tensor (what the fused HSTU layer hands over) and
    scaling_seqlen decoupled from max_seqlen."""
    rng = np.random.default_rng(3)
    H, d = 4, 64
    lengths = np.array([70, 129, 5])
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    fused = torch.empty(T, 3, H, d, device=DEV).uniform_(-1, 1).bfloat16()
    q, k, v = fused[:, 0], fused[:, 1], fused[:, 2]
    out, _ = _run(q, k, v, off, 129, None, None, 1, True, 0.125, scaling=1000)
    ref = ho.hstu_attn_fwd(*(t.float().cpu().numpy() for t in (q, k, v)), off, 0.125, 1000, True)
    assert np.abs(out.detach().float().cpu().numpy() - ref).max() <= 6e-3 * np.abs(ref).max()


def test_rejects_unsupported():
    from hstu import hstu_attn_varlen_func

    q = torch.zeros(4, 1, 48, device=DEV, dtype=torch.bfloat16)
    cu = torch.tensor([0, 4], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        hstu_attn_varlen_func(q, q, q, cu, cu, None, None, 4, 4, 4, None, None)  # head_dim 48
    q = torch.zeros(4, 1, 32, device=DEV, dtype=torch.bfloat16)
    nt = torch.tensor([1], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        hstu_attn_varlen_func(q, q, q, cu, cu, None, None, 4, 4, 4, None, nt, window_size=(-1, -1))  # targets need causal
    with pytest.raises(RuntimeError):
        hstu_attn_varlen_func(q.float(), q.float(), q.float(), cu, cu, None, None, 4, 4, 4, None, None)
