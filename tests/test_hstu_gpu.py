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


def _run(q, k, v, off, N, targets, ctx, grp, causal, alpha, dout=None, scaling=None, window=None):
    from hstu import hstu_attn_varlen_func

    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    cu = torch.from_numpy(np.asarray(off, np.int32)).to(DEV)
    nt = None if targets is None else torch.from_numpy(np.asarray(targets, np.int32)).to(DEV)
    nc = None if ctx is None else torch.from_numpy(np.asarray(ctx, np.int32)).to(DEV)
    out = hstu_attn_varlen_func(qq, kk, vv, cu, cu, None, None, N, N, scaling if scaling is not None else N, nc, nt,
                                target_group_size=grp,
                                window_size=window if window is not None else ((-1, 0) if causal else (-1, -1)), alpha=alpha)
    if dout is None:
        return out, None
    out.backward(dout)
    return out, (qq.grad, kk.grad, vv.grad)


def _bf16_ulp(x):
    """spacing of bfloat16 at |x| (8 significand bits)"""
    ax = np.maximum(np.abs(x), 2.0 ** -126)
    return 2.0 ** (np.floor(np.log2(ax)) - 7)


def _close_elementwise(actual, ref, mag, k, bits=7):
    """(bits = fraction bits of the operand type: 7 bf16, 10 fp16 -- ulp = 2^(e - bits), rounding floor k * 2^-(bits + 2).)
    The bf16 rule of path A (tests/test_demb_gpu.py: assert_close_lowp), element by element: |x - ref| <= 1e-3 |ref| +
    1 ulp_bf16(ref) + k * 2^-9 * mag, where mag = the accumulated magnitude of the element's summands (oracle:
    hstu_attn_magnitudes) and k * 2^-9 the relative rounding the kernels apply to them before the second GEMMs (P: one bf16
    rounding, k = 2 with margin; dS: P-dependent products of two rounded factors, k = 4).  The floor is per ELEMENT, so an
    error confined to small rows cannot hide under the tensor's maximum."""
    a = actual.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    ulp = _bf16_ulp(ref) * 2.0 ** (7 - bits)
    if bits == 10:
        ulp = np.maximum(ulp, 2.0 ** -24)       # fp16 is subnormal below 6.1e-5: fixed spacing (the first tokens' outputs, ~ 1 / N)
    tol = 1e-3 * np.abs(ref) + ulp + k * 2.0 ** -(bits + 2) * np.asarray(mag, np.float64) + 1e-30
    ratio = float((np.abs(a - ref) / tol).max()) if a.size else 0.0
    try:   # the worst use of the tolerance goes to the terminal summary (conftest.py)
        from conftest import record_tolerance_use

        record_tolerance_use(f"{'fp16' if bits == 10 else 'bf16'} k={k}", os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], ratio)
    except ImportError:
        pass
    bad = np.abs(a - ref) > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} elements off: worst excess {ratio:.2f} x its tolerance"


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
    _close(out, g("out_bf16"), g("out"), 2)          # the reference's own rule (2x / 5x the error of its bf16 run), and
    _close(grads[0], g("dq_bf16"), g("dq"), 5)
    _close(grads[1], g("dk_bf16"), g("dk"), 5)
    _close(grads[2], g("dv_bf16"), g("dv"), 5)
    # BASELINE.json's bound element by element: 1e-3 relative + 1 ulp(bf16) + a floor tied to the element's accumulated magnitude
    qn, kn, vn, dn = (np.asarray(_bf(g(x)).float().cpu()) for x in ("q", "k", "v", "dout"))
    mo, mq, mk, mv = ho.hstu_attn_magnitudes(dn, qn, kn, vn, g("off"), 1.0 / d ** 0.5, N, bool(causal), targets, ctx, grp)
    _close_elementwise(out, g("out"), mo, 2)
    _close_elementwise(grads[0], g("dq"), mq, 4)
    _close_elementwise(grads[1], g("dk"), mk, 4)
    _close_elementwise(grads[2], g("dv"), mv, 4)


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
    mo, mq, mk, mv = ho.hstu_attn_magnitudes(dn, qn, kn, vn, off, alpha, maxL, causal, targets, ctx, grp)
    for got, want, mag, kk in ((out, ref, mo, 2), (grads[0], dq, mq, 4), (grads[1], dk, mk, 4), (grads[2], dv, mv, 4)):
        _close_elementwise(got, want, mag, kk)


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("mode", ["causal", "ctx_targets", "group_targets", "noncausal"])
def test_backward_exchange_is_bit_identical_to_the_recomputing_passes(d, mode, monkeypatch):
    """the dK pass leaves dS (d >= 128: and P) in a scratch buffer and dV / dQ are one-GEMM passes over them
    (hstu_bwd_v_p_kernel / hstu_bwd_q_ds_kernel); without the buffer three passes recompute S.  Same bf16 operands in the
    same GEMMs: dq, dk, dv must agree bit for bit -- over several key / query blocks, ragged ends, empty and 1-token
    sequences, targets (with and without groups) and contextual rows, causal and not."""
    import hstu.hstu_attn_interface as hi

    rng = np.random.default_rng(d + len(mode))
    lengths = np.array([700, 1, 0, 333, 129, 64, 257, 31])
    off = np.concatenate([[0], np.cumsum(lengths)])
    T, H = int(off[-1]), 2
    causal = mode != "noncausal"
    targets = ctx = None
    grp = 1
    if mode in ("ctx_targets", "group_targets"):
        targets = np.minimum(rng.integers(0, 40, lengths.size), lengths)
        ctx = np.minimum(rng.integers(0, 20, lengths.size), lengths - targets)
        grp = 3 if mode == "group_targets" else 1
    mk = lambda: torch.from_numpy(rng.uniform(-1, 1, (T, H, d)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    N = int(lengths.max())
    assert hi.lib().mi355_hstu_attn_bwd_ds_bytes(lengths.size, H, d, N) > 0
    _, g_x = _run(q, k, v, off, N, targets, ctx, grp, causal, 1.0 / d ** 0.5, dout=dout)
    monkeypatch.setattr(hi, "_DS_MAX_BYTES", 0)          # no scratch buffer: the recomputing passes
    _, g_r = _run(q, k, v, off, N, targets, ctx, grp, causal, 1.0 / d ** 0.5, dout=dout)
    for a, b, name in zip(g_x, g_r, ("dq", "dk", "dv")):
        assert torch.equal(a, b), f"{name}: {(a.float() - b.float()).abs().max().item()}"
    assert float(g_x[0].float().abs().max()) > 0


@pytest.mark.parametrize("d", [64, 256])
@pytest.mark.parametrize("mode", ["causal", "ctx_targets", "noncausal"])
def test_backward_exchange_in_chunks_under_a_cap_is_bit_identical(d, mode, monkeypatch):
    """MI355_HSTU_DS_MAX_BYTES caps the P / dS scratch: above it the buffer holds one chunk of (sequence, head) units at a
    time -- each with the tiles of its own length (jagged layout, planned on the device) -- and the three passes run once per
    chunk.  Same kernels, same tiles: dq, dk, dv agree bit for bit with the dense one-pass exchange.  The cap here leaves
    room for two units of the longest sequence, so the 16 units of the batch take many chunks."""
    import hstu.hstu_attn_interface as hi

    rng = np.random.default_rng(7 * d + len(mode))
    lengths = np.array([700, 1, 0, 333, 129, 64, 257, 31])
    off = np.concatenate([[0], np.cumsum(lengths)])
    T, H = int(off[-1]), 2
    causal = mode != "noncausal"
    targets = ctx = None
    if mode == "ctx_targets":
        targets = np.minimum(rng.integers(0, 40, lengths.size), lengths)
        ctx = np.minimum(rng.integers(0, 20, lengths.size), lengths - targets)
    mk = lambda: torch.from_numpy(rng.uniform(-1, 1, (T, H, d)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    N = int(lengths.max())
    L = hi.lib()
    dense = L.mi355_hstu_attn_bwd_ds_bytes(lengths.size, H, d, N)
    assert dense > 0
    _, g_dense = _run(q, k, v, off, N, targets, ctx, 1, causal, 1.0 / d ** 0.5, dout=dout)
    ng = (N + 31) // 32
    tri = causal and ctx is None
    umax = ng * (ng + 1) // 2 if tri else ng * ng
    regions = 2 if d >= 128 else 1
    cap = 4096 + regions * int(2.3 * umax) * 2048
    assert cap < dense
    got = L.mi355_hstu_attn_bwd_ds_bytes_capped(lengths.size, H, d, N, T, cap, int(tri))
    assert 0 < got <= cap
    # below two units of the longest sequence nothing fits: the library says so (0) and the recomputing passes run
    assert L.mi355_hstu_attn_bwd_ds_bytes_capped(lengths.size, H, d, N, T, 4096 + regions * int(1.5 * umax) * 2048, int(tri)) == 0
    monkeypatch.setattr(hi, "_DS_MAX_BYTES", cap)
    _, g_chunk = _run(q, k, v, off, N, targets, ctx, 1, causal, 1.0 / d ** 0.5, dout=dout)
    for a, b, name in zip(g_chunk, g_dense, ("dq", "dk", "dv")):
        assert torch.equal(a, b), f"{name}: {(a.float() - b.float()).abs().max().item()}"
    assert float(g_chunk[0].float().abs().max()) > 0


def test_backward_scratch_stays_under_one_gib_at_32x4096(monkeypatch):
    """The dense layout of the exchange would take B H ceil(L / 32)^2 x 4 KB = 8.6 GB at 32 sequences x 4096 tokens x 4 heads
    (d = 256); under a cap of 1 GiB (MI355_HSTU_DS_MAX_BYTES; the default is 4 GiB, which is 19 % faster here) the backward
    allocates at most 1 GiB of scratch next to its three outputs (peak allocator use checked), and its dq, dk, dv are those of
    the recomputing passes bit for bit."""
    import hstu.hstu_attn_interface as hi

    Bq, L, H, d = 32, 4096, 4, 256
    T = Bq * L
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    q, k, v, dout = (torch.empty(T, H, d, device=DEV).uniform_(-1, 1, generator=g).bfloat16() for _ in range(4))
    assert hi.lib().mi355_hstu_attn_bwd_ds_bytes(Bq, H, d, L) >= (8 << 30)
    assert hi._DS_MAX_BYTES <= (4 << 30) or "MI355_HSTU_DS_MAX_BYTES" in os.environ      # the default never asks for the dense 8.6 GB
    monkeypatch.setattr(hi, "_DS_MAX_BYTES", 1 << 30)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    dq, dk, dv = hi.hstu_varlen_bwd(dout, q, k, v, cu, L, L, None, None, 1, True, 1.0 / d ** 0.5)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    outs = 3 * T * H * d * 2
    assert peak <= outs + (1 << 30) + (64 << 20), f"backward peak {peak >> 20} MB for {outs >> 20} MB of outputs"
    monkeypatch.setattr(hi, "_DS_MAX_BYTES", 4 << 30)     # the default: two chunks
    dq4, dk4, dv4 = hi.hstu_varlen_bwd(dout, q, k, v, cu, L, L, None, None, 1, True, 1.0 / d ** 0.5)
    assert torch.equal(dq, dq4) and torch.equal(dk, dk4) and torch.equal(dv, dv4)
    monkeypatch.setattr(hi, "_DS_MAX_BYTES", 0)           # no scratch at all: the recomputing passes
    rq, rk, rv = hi.hstu_varlen_bwd(dout, q, k, v, cu, L, L, None, None, 1, True, 1.0 / d ** 0.5)
    assert torch.equal(dq, rq) and torch.equal(dk, rk) and torch.equal(dv, rv)


# ------------------------------------------------------------------------------------------ local (sliding) windows
W = np.load(os.path.join(os.path.dirname(__file__), "golden", "hstu_window_golden.npz"))
WINDOWS = [(111, 11), (111, 222), (50, 0), (0, 0), (0, 7), (-1, 40), (64, -1), (1000, 0), (3, 1000), (200, 130)]


def _assert_vs_oracle(out, grads, ref, dq, dk, dv, mags=None):
    """element-wise (mags = hstu_attn_magnitudes of the same call) -- or, for the callers that cannot restate the magnitudes,
    the max-norm form"""
    if mags is not None:
        for got, want, mag, kk in ((out, ref, mags[0], 2), (grads[0], dq, mags[1], 4), (grads[1], dk, mags[2], 4), (grads[2], dv, mags[3], 4)):
            _close_elementwise(got, want, mag, kk)
        return
    for got, want, tol in ((out, ref, 6e-3), (grads[0], dq, 1.2e-2), (grads[1], dk, 1.2e-2), (grads[2], dv, 1.2e-2)):
        gn = got.detach().float().cpu().numpy()
        err = np.abs(gn - want).max()
        assert err <= tol * np.abs(want).max() + 1e-6, f"{err} vs scale {np.abs(want).max()}"


@pytest.mark.parametrize("name", [str(c) for c in W["cases"]])
def test_local_window_golden(name):
    """sliding window forward + backward against the reference test's own dense statement (construct_mask +
    _hstu_attention_maybe_from_cache of corelib/hstu/test.py, fp32; tests/golden/gen_hstu_window_golden.py)."""
    H, d, wl, wr, N = (int(x) for x in W[f"{name}/meta"])
    g = lambda k: W[f"{name}/{k}"]
    out, grads = _run(_bf(g("q")), _bf(g("k")), _bf(g("v")), g("off"), N, None, None, 1, False, 1.0 / d ** 0.5,
                      dout=_bf(g("dout")), window=(wl, wr))
    qn, kn, vn, dn = (np.asarray(_bf(g(x)).float().cpu()) for x in ("q", "k", "v", "dout"))
    mags = ho.hstu_attn_magnitudes(dn, qn, kn, vn, g("off"), 1.0 / d ** 0.5, N, local_window=(wl, wr))
    _assert_vs_oracle(out, grads, g("out"), g("dq"), g("dk"), g("dv"), mags)


def test_local_window_with_the_mask_alone():
    """by default the tile loops are clipped to the window's band; MI355_HSTU_WSKIP=0 keeps the full loops and lets the
    per-element mask do all the work.  The library reads the switch once, so the window tests are re-run in a child
    process with it: same goldens, same oracle comparisons, same exchange bit-identity."""
    import subprocess
    import sys

    if os.environ.get("MI355_HSTU_WSKIP") == "0":
        pytest.skip("already the unclipped run")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "local_window and not mask_alone"], env=dict(os.environ, MI355_HSTU_WSKIP="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("window", WINDOWS, ids=lambda w: f"w{w[0]}_{w[1]}")
def test_local_window_random_jagged_vs_oracle(d, window):
    """windows narrower and wider than the 64 / 128-row tiles, one-sided, zero-width sides and wider than the sequence
    (hstu_api.cpp:154-165: a side < 0 or > max_seqlen_k is unbounded), over sequences of several tiles, ragged ends,
    an empty and a 1-token sequence."""
    rng = np.random.default_rng(d + 7 * window[0] + window[1])
    lengths = np.array([517, 1, 0, 333, 129, 64])
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T, H, N = int(off[-1]), 2, int(lengths.max())
    mk = lambda lo, hi: torch.from_numpy(rng.uniform(lo, hi, (T, H, d)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(-1, 1), mk(-1, 1), mk(-1, 1), mk(0, 1)
    alpha = 1.0 / d ** 0.5
    out, grads = _run(q, k, v, off, N, None, None, 1, False, alpha, dout=dout, window=window)
    qn, kn, vn, dn = (t.float().cpu().numpy() for t in (q, k, v, dout))
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, N, local_window=window)
    dq, dk, dv = ho.hstu_attn_bwd(dn, qn, kn, vn, off, alpha, N, local_window=window)
    _assert_vs_oracle(out, grads, ref, dq, dk, dv, ho.hstu_attn_magnitudes(dn, qn, kn, vn, off, alpha, N, local_window=window))


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("window", [(70, 0), (33, 200), (-1, 65), (300, -1)], ids=lambda w: f"w{w[0]}_{w[1]}")
def test_local_window_exchange_is_bit_identical_to_the_recomputing_passes(d, window, monkeypatch):
    """the dV / dQ passes replay the dK pass' clipped query span (kv_span) to know which exchanged sub-tiles exist: with
    and without the scratch buffer the gradients must agree bit for bit, and the clipped loops (default) must agree bit
    for bit with... the forward of the unclipped ones is covered by test_local_window_golden[*-0]."""
    import hstu.hstu_attn_interface as hi

    rng = np.random.default_rng(d + window[0] + window[1])
    lengths = np.array([700, 1, 0, 333, 129, 64, 257, 31])
    off = np.concatenate([[0], np.cumsum(lengths)])
    T, H, N = int(off[-1]), 2, int(lengths.max())
    mk = lambda: torch.from_numpy(rng.uniform(-1, 1, (T, H, d)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    o_x, g_x = _run(q, k, v, off, N, None, None, 1, False, 1.0 / d ** 0.5, dout=dout, window=window)
    monkeypatch.setattr(hi, "_DS_MAX_BYTES", 0)
    o_r, g_r = _run(q, k, v, off, N, None, None, 1, False, 1.0 / d ** 0.5, dout=dout, window=window)
    assert torch.equal(o_x, o_r)
    for a, b, name in zip(g_x, g_r, ("dq", "dk", "dv")):
        assert torch.equal(a, b), f"{name}: {(a.float() - b.float()).abs().max().item()}"
    assert float(g_x[0].float().abs().max()) > 0


def test_local_window_degenerate_pairs_equal_the_plain_masks():
    """(L+, 0) is the causal mask and (L+, L+) the full one, bit for bit (hstu_api.cpp:154-159)."""
    rng = np.random.default_rng(11)
    lengths = np.array([300, 77, 129])
    off = np.concatenate([[0], np.cumsum(lengths)])
    T, H, d, N = int(off[-1]), 2, 64, 300
    mk = lambda: torch.from_numpy(rng.uniform(-1, 1, (T, H, d)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    for window, causal in (((299, 0), True), ((5000, 0), True), ((299, 299), False), ((-1, 299), False), ((299, -1), False)):
        o_w, g_w = _run(q, k, v, off, N, None, None, 1, False, 0.125, dout=dout, window=window)
        o_p, g_p = _run(q, k, v, off, N, None, None, 1, causal, 0.125, dout=dout)
        assert torch.equal(o_w, o_p), window
        for a, b in zip(g_w, g_p):
            assert torch.equal(a, b), window


def test_local_window_rejects_contexts_and_targets():
    from hstu import hstu_attn_varlen_func

    q = torch.zeros(8, 1, 32, device=DEV, dtype=torch.bfloat16)
    cu = torch.tensor([0, 8], dtype=torch.int32, device=DEV)
    one = torch.tensor([1], dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError):   # hstu_attn_interface.py:238-245 of the reference
        hstu_attn_varlen_func(q, q, q, cu, cu, None, None, 8, 8, 8, one, None, window_size=(3, 0))
    with pytest.raises(ValueError):
        hstu_attn_varlen_func(q, q, q, cu, cu, None, None, 8, 8, 8, None, one, window_size=(3, 2))


# ------------------------------------------------------------------------------------- relative attention bias (rab)
R = np.load(os.path.join(os.path.dirname(__file__), "golden", "hstu_rab_golden.npz"))


def _run_rab(q, k, v, rab, off, N, targets, ctx, grp, window, alpha, dout, has_drab=True):
    from hstu import hstu_attn_varlen_func

    qq, kk, vv, rr = (t.clone().requires_grad_(True) for t in (q, k, v, rab))
    cu = torch.from_numpy(np.asarray(off, np.int32)).to(DEV)
    nt = None if targets is None else torch.from_numpy(np.asarray(targets, np.int32)).to(DEV)
    nc = None if ctx is None else torch.from_numpy(np.asarray(ctx, np.int32)).to(DEV)
    out = hstu_attn_varlen_func(qq, kk, vv, cu, cu, None, None, N, N, N, nc, nt, target_group_size=grp, window_size=window,
                                alpha=alpha, rab=rr, has_drab=has_drab)
    out.backward(dout)
    return out, (qq.grad, kk.grad, vv.grad), rr.grad


def _assert_drab(got, want):
    gn = got.detach().float().cpu().numpy()
    assert gn.shape == want.shape
    err = np.abs(gn - want).max()
    assert err <= 1.2e-2 * np.abs(want).max() + 1e-6, f"drab: {err} vs scale {np.abs(want).max()}"


@pytest.mark.parametrize("name", [str(c) for c in R["cases"]])
def test_rab_golden(name):
    """rab forward + backward + drab against the reference test's dense statement (tests/golden/gen_hstu_rab_golden.py)"""
    H, HR, d, wl, wr, N = (int(x) for x in R[f"{name}/meta"])
    g = lambda k: R[f"{name}/{k}"]
    out, grads, drab = _run_rab(_bf(g("q")), _bf(g("k")), _bf(g("v")), _bf(g("rab")), g("off"), N, None, None, 1, (wl, wr),
                                1.0 / d ** 0.5, _bf(g("dout")))
    _assert_vs_oracle(out, grads, g("out"), g("dq"), g("dk"), g("dv"))
    _assert_drab(drab, g("drab"))


# ------------------------------------------------------------------------------------- arbitrary mask functions (func)
FN = np.load(os.path.join(os.path.dirname(__file__), "golden", "hstu_func_golden.npz"))


@pytest.mark.parametrize("name", [str(c) for c in FN["cases"]])
def test_arbitrary_mask_golden(name):
    """`func` (hstu_api.cpp:170-180): forward and dq / dk / dv against the reference test's dense statement
    (tests/golden/gen_hstu_func_golden.py: its random three- / five-bound functions, its causal and local-window emulations)"""
    from hstu import hstu_attn_varlen_func

    H, d, N = (int(x) for x in FN[f"{name}/meta"])
    g = lambda k: FN[f"{name}/{k}"]
    qq, kk, vv = (_bf(g(t)).clone().requires_grad_(True) for t in ("q", "k", "v"))
    cu = torch.from_numpy(g("off").astype(np.int32)).to(DEV)
    func = torch.from_numpy(g("func")).to(DEV)
    out = hstu_attn_varlen_func(qq, kk, vv, cu, cu, None, None, N, N, N, None, None, window_size=(-1, -1), alpha=1.0 / d ** 0.5,
                                func=func)
    out.backward(_bf(g("dout")))
    _assert_vs_oracle(out, (qq.grad, kk.grad, vv.grad), g("out"), g("dq"), g("dk"), g("dv"))


@pytest.mark.parametrize("mode", ["per_head", "with_rab", "with_targets_causal", "delta_q", "delta_q_d256", "per_head_d256", "raw_ops", "fp16",
                                  "fp16_with_rab"])
def test_arbitrary_mask_random_jagged_vs_oracle(mode):
    """func over several tiles with ragged ends, an empty and a one-token sequence: one mask per head; together with a relative
    bias (drab flows through the sum, zero where the function masks); narrowing the causal + target mask; over delta-q keys
    (inference forward: the queries are the last rows of their sequences); through the raw fbgemm ops of the fused layer."""
    from hstu import hstu_attn_varlen_func

    rng = np.random.default_rng(len(mode))
    lengths = np.array([300, 1, 0, 129, 64, 77])
    B, H, d, N = lengths.size, 2, (256 if mode.endswith("_d256") else 64), int(lengths.max())   # (d = 256: the two-wave-kind forward, the exchange backward)
    mode = mode[:-5] if mode.endswith("_d256") else mode
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    tdt = torch.float16 if mode.startswith("fp16") else torch.bfloat16      # (fp16: the mask bias must stay finite, -6e4)
    mk = lambda lo, hi, *shape: torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32)).to(DEV).to(tdt)
    q, k, v, dout = mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(0, 1, T, H, d)
    HF = H if mode == "per_head" else 1
    f = np.zeros((HF, 5, T + 64), np.int32)
    f[:, 0] = rng.integers(0, 40, size=(HF, T + 64))
    f[:, 1] = rng.integers(30, 120, size=(HF, T + 64)); f[:, 2] = f[:, 1] + rng.integers(0, 90, size=(HF, T + 64))
    f[:, 3] = rng.integers(200, 260, size=(HF, T + 64)); f[:, 4] = f[:, 3] + rng.integers(0, 60, size=(HF, T + 64))
    func = torch.from_numpy(f).to(DEV)
    cu = torch.from_numpy(off.astype(np.int32)).to(DEV)
    alpha = 1.0 / d ** 0.5
    qn, kn, vn, dn = (t.float().cpu().numpy() for t in (q, k, v, dout))
    if mode == "delta_q":
        # the last min(L, 37) rows of every sequence are the queries; keys = the whole sequence
        lq = np.minimum(lengths, 37)
        offq = np.concatenate([[0], np.cumsum(lq)]).astype(np.int64)
        qsel = np.concatenate([np.arange(off[b + 1] - lq[b], off[b + 1]) for b in range(B)]).astype(np.int64)
        qd = q[torch.from_numpy(qsel).to(DEV)].contiguous()
        fd = np.zeros((1, 5, int(offq[-1]) + 8), np.int32)
        fd[:, :, : int(offq[-1])] = f[:, :, qsel]
        cuq = torch.from_numpy(offq.astype(np.int32)).to(DEV)
        with torch.no_grad():
            out = hstu_attn_varlen_func(qd, k, v, cuq, cu, None, None, int(lq.max()), N, N, None, None, window_size=(-1, -1),
                                        alpha=alpha, func=torch.from_numpy(fd).to(DEV))
        full = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, N, causal=False, func=f)    # row r of the sequence = its token: same mask rows
        ref = full[qsel]
        err = np.abs(out.float().cpu().numpy() - ref).max()
        assert err <= 6e-3 * np.abs(ref).max() + 1e-6, f"{err} vs scale {np.abs(ref).max()}"
        return
    kw, okw = {}, dict(causal=False)
    rab = None
    if mode in ("with_rab", "fp16_with_rab"):
        rab = mk(-2, 2, B, H, N, N).requires_grad_(True)
        kw.update(rab=rab, has_drab=True)
        okw.update(rab=rab.detach().float().cpu().numpy())
    targets = None
    if mode == "with_targets_causal":
        targets = np.minimum(rng.integers(0, 11, size=B), np.maximum(lengths - 1, 0))
        kw.update(window_size=(-1, 0), target_group_size=2)
        okw.update(causal=True, num_targets=targets, target_group_size=2)
    else:
        kw.update(window_size=(-1, -1))
    nt = None if targets is None else torch.from_numpy(targets.astype(np.int32)).to(DEV)
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    if mode == "raw_ops":
        import hstu.hstu_ops_gpu  # noqa: F401  (registers torch.ops.fbgemm.hstu_varlen_*)
        out, _ = torch.ops.fbgemm.hstu_varlen_fwd_80(q, k, v, cu, cu, None, None, N, N, float(N), None, None, 1, -1, -1, alpha, None, func)
        grads = torch.ops.fbgemm.hstu_varlen_bwd_80(dout, q, k, v, cu, cu, None, None, N, N, float(N), None, None, None, None, None, 1,
                                                    -1, -1, alpha, None, False, func, False)[:3]
    else:
        out = hstu_attn_varlen_func(qq, kk, vv, cu, cu, None, None, N, N, N, None, nt, alpha=alpha, func=func, **kw)
        out.backward(dout)
        grads = (qq.grad, kk.grad, vv.grad)
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, N, func=f, **okw)
    res = ho.hstu_attn_bwd(dn, qn, kn, vn, off, alpha, N, func=f, **okw)
    _assert_vs_oracle(out, grads, ref, res[0], res[1], res[2])
    assert bool(torch.isfinite(out.float()).all()) and all(bool(torch.isfinite(g_.float()).all()) for g_ in grads)
    if mode in ("with_rab", "fp16_with_rab"):
        _assert_drab(rab.grad, res[3])


def _func_case(rng, lengths, H, d, tdt, per_head=True, slack=64):
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    mk = lambda lo, hi, *shape: torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32)).to(DEV).to(tdt)
    q, k, v, dout = mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(0, 1, T, H, d)
    HF = H if per_head else 1
    f = np.zeros((HF, 5, T + slack), np.int32)
    f[:, 0] = rng.integers(0, 40, size=(HF, T + slack))
    f[:, 1] = rng.integers(30, 120, size=(HF, T + slack)); f[:, 2] = f[:, 1] + rng.integers(0, 90, size=(HF, T + slack))
    f[:, 3] = rng.integers(200, 260, size=(HF, T + slack)); f[:, 4] = f[:, 3] + rng.integers(0, 60, size=(HF, T + slack))
    return off, q, k, v, dout, f


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [64, 256])
def test_mask_functions_inside_the_kernels_match_their_dense_bias_statement_bit_for_bit(tdt, d):
    """round 5: `func` is read inside the forward / dK+dV / dQ kernels (mi355_hstu_attn_{fwd_kv,bwd}_func) instead of being expanded
    into a [batch, heads, N, N] bias of 0 / -1e9 for the biased kernels.  The two statements of the mask agree bit for bit:
    output, dq, dk, dv; full, causal + targets, and a local window on top."""
    from hstu.hstu_attn_interface import HstuAttnFuncFunc, HstuAttnRabFunc, func_mask_bias

    rng = np.random.default_rng(7 + d)
    lengths = np.array([300, 1, 0, 129, 64, 77])
    off, q, k, v, dout, f = _func_case(rng, lengths, 2, d, tdt)
    B, N = lengths.size, int(lengths.max())
    cu = torch.from_numpy(off.astype(np.int32)).to(DEV)
    func = torch.from_numpy(f).to(DEV)
    nt = torch.from_numpy(np.minimum(rng.integers(0, 11, size=B), np.maximum(lengths - 1, 0)).astype(np.int32)).to(DEV)
    alpha = 1.0 / d ** 0.5
    for wl, wr, targets in ((-1, -1, None), (-1, 0, nt), (40, 25, None)):
        res = []
        for kind in ("kernel", "dense"):
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            if kind == "kernel":
                out = HstuAttnFuncFunc.apply(qq, kk, vv, func, cu, N, float(N), None, targets, 2, wl, wr, alpha)
            else:
                out = HstuAttnRabFunc.apply(qq, kk, vv, func_mask_bias(func, cu, cu, N, tdt), cu, N, float(N), None, targets, 2, wl,
                                            wr, alpha, False)
            out.backward(dout)
            res.append((out.detach(), qq.grad, kk.grad, vv.grad))
        for a_, b_ in zip(*res):
            assert torch.equal(a_, b_), f"window ({wl}, {wr}): the in-kernel mask functions differ from the dense bias"


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", ["prefix_gap_band", "causal_like", "bands_only", "three_bounds", "one_bound", "blind_rows", "dense_pairs", "sink_window",
                                   "sink_window_dense"])
def test_mask_functions_on_the_two_wave_kind_forward_match_the_dense_bias_statement(shape, tdt):
    """round 6: at head dim 256 functions of up to two bands ride the 64-rows-per-wave forward (hstu_fwd_q2_kernel<.., kFunc>): tile
    stream clipped to the block's extents and jumping over the tiles between the block's prefix and its bands, a half skipping what is
    left of its own gap, mask-free tiles below every row's prefix; the backward on the P / dS exchange, its dQ pass with the same jump.  Against the dense 0 / -1e9 bias statement of the same mask through the one-kind kernel, bit for bit
    (forward and the three gradients), over sequences of several row blocks with ragged ends."""
    from hstu.hstu_attn_interface import HstuAttnFuncFunc, HstuAttnRabFunc, func_mask_bias

    rng = np.random.default_rng(len(shape))
    lengths = np.array([1024, 1024, 1024]) if shape in ("dense_pairs", "sink_window_dense") else np.array([1500, 0, 700, 1, 130, 257])
    H, d = 2, 256
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T, B, N = int(off[-1]), lengths.size, int(lengths.max())
    mk = lambda lo, hi, *sh: torch.from_numpy(rng.uniform(lo, hi, sh).astype(np.float32)).to(DEV).to(tdt)
    q, k, v, dout = mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(0, 1, T, H, d)
    pos = np.concatenate([np.arange(n) for n in lengths]).astype(np.int64)
    nb = {"three_bounds": 3, "one_bound": 1}.get(shape, 5)
    f = np.zeros((H, nb, T), np.int64)
    r = lambda lo, hi: rng.integers(lo, hi, size=(H, T))
    if shape == "prefix_gap_band":            # a short prefix, two bands far to the right: whole tiles between them see nothing
        f[:, 0] = r(0, 100); f[:, 1] = r(900, 1000); f[:, 2] = f[:, 1] + r(0, 200); f[:, 3] = r(1250, 1300); f[:, 4] = f[:, 3] + r(0, 150)
    elif shape in ("causal_like", "dense_pairs"):   # j <= i and a band inside it, a second one to the right of the diagonal
        f[:, 0] = pos + 1; f[:, 1] = np.maximum(pos - 300, 0); f[:, 2] = np.maximum(pos - 200, 0); f[:, 3] = pos + 100; f[:, 4] = pos + 140
    elif shape == "bands_only":               # no prefix: a sliding window and a sink, as bands
        f[:, 1] = 0; f[:, 2] = r(0, 20); f[:, 3] = np.maximum(pos - r(100, 200), 0); f[:, 4] = pos + r(0, 3)
    elif shape in ("sink_window", "sink_window_dense"):   # the first keys and a causal window: whole tiles between them leave the block's stream (the gap jump)
        f[:, 0] = np.minimum(pos + 1, r(40, 70)); f[:, 1] = np.maximum(pos - r(150, 260), 0); f[:, 2] = pos + 1; f[:, 3] = f[:, 4] = 0
    elif shape == "three_bounds":
        f[:, 0] = r(0, 300); f[:, 1] = r(400, 600); f[:, 2] = f[:, 1] + r(0, 500)
    elif shape == "one_bound":
        f[:, 0] = np.where(r(0, 4) == 0, 0, pos + 1 - r(0, 2))
    elif shape == "blind_rows":               # half of the 64-row groups see nothing at all, the others a band
        blind = ((pos // 64) % 2 == 0)[None, :]
        f[:, 1] = np.where(blind, 0, r(500, 520)); f[:, 2] = np.where(blind, 0, f[:, 1] + r(1, 90)); f[:, 3] = f[:, 4] = 0
    func = torch.from_numpy(f.astype(np.int32)).to(DEV)
    cu = torch.from_numpy(off.astype(np.int32)).to(DEV)
    alpha = 1.0 / d ** 0.5
    for wl, wr in ((-1, -1), (-1, 0), (450, 60)):
        res = []
        for kind in ("kernel", "dense"):
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            if kind == "kernel":
                out = HstuAttnFuncFunc.apply(qq, kk, vv, func, cu, N, float(N), None, None, 1, wl, wr, alpha)
            else:
                out = HstuAttnRabFunc.apply(qq, kk, vv, func_mask_bias(func, cu, cu, N, tdt), cu, N, float(N), None, None, 1, wl, wr, alpha, False)
            out.backward(dout)
            res.append((out.detach(), qq.grad, kk.grad, vv.grad))
        for name, a_, b_ in zip(("out", "dq", "dk", "dv"), *res):
            assert torch.equal(a_, b_), f"{shape}, window ({wl}, {wr}): {name} differs from the dense bias statement"
        assert float(res[0][0].float().abs().sum()) > 0


@pytest.mark.parametrize("mode", ["full", "causal_targets", "contexts", "window"])
def test_mask_functions_on_the_exchange_backward_match_the_recomputing_passes_bit_for_bit(mode, monkeypatch):
    """round 6: at head dim 256 functions of up to two bands take the P / dS exchange backward (hstu_bwd_kv_pc_kernel<.., kFunc> writes
    the sub-tiles of the steps its key block's table entry names, the one-GEMM passes replay the entry).  dq, dk, dv agree bit for bit
    with the recomputing passes (no scratch: _DS_MAX_BYTES = 0) and with the jagged, chunked layout of the exchange under a cap."""
    import hstu.hstu_attn_interface as hi
    from hstu.hstu_attn_interface import HstuAttnFuncFunc

    rng = np.random.default_rng(len(mode) + 3)
    lengths = np.array([700, 1, 0, 333, 129, 64, 257, 31])
    H, d = 2, 256
    off, q, k, v, dout, f = _func_case(rng, lengths, H, d, torch.bfloat16)
    T, B, N = int(off[-1]), lengths.size, int(lengths.max())
    pos = np.concatenate([np.arange(n) for n in lengths])
    f[:, 0, :T] = np.where(rng.integers(0, 3, size=(H, T)) == 0, pos + 1, f[:, 0, :T])      # a third of the rows: a causal prefix
    f[:, 3, :T] = rng.integers(400, 500, size=(H, T)); f[:, 4, :T] = f[:, 3, :T] + rng.integers(0, 200, size=(H, T))
    cu = torch.from_numpy(off.astype(np.int32)).to(DEV)
    func = torch.from_numpy(f).to(DEV)
    ti = lambda a: torch.from_numpy(a.astype(np.int32)).to(DEV)
    ctx = tgt = None
    wl, wr = -1, -1
    if mode == "causal_targets":
        tgt = ti(np.minimum(rng.integers(0, 30, size=B), np.maximum(lengths - 1, 0))); wr = 0
    elif mode == "contexts":
        cx = np.minimum(rng.integers(0, 12, size=B), lengths)
        ctx = ti(cx); tgt = ti(np.minimum(rng.integers(0, 20, size=B), lengths - cx)); wr = 0
    elif mode == "window":
        wl, wr = 350, 40
    alpha = 1.0 / d ** 0.5

    def grads():
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = HstuAttnFuncFunc.apply(qq, kk, vv, func, cu, N, float(N), ctx, tgt, 2, wl, wr, alpha)
        out.backward(dout)
        return out.detach(), qq.grad, kk.grad, vv.grad

    ref = None
    ng = (N + 31) // 32
    cap = 4096 + 2 * int(2.3 * ng * ng) * 2048
    assert cap < hi.lib().mi355_hstu_attn_bwd_ds_bytes(B, H, d, N)
    for label, limit in (("recompute", 0), ("dense exchange", 4 << 30), ("chunked exchange", cap)):
        monkeypatch.setattr(hi, "_DS_MAX_BYTES", limit)
        got = grads()
        if ref is None:
            ref = got
            assert float(ref[1].float().abs().max()) > 0
            continue
        for name, a_, b_ in zip(("out", "dq", "dk", "dv"), got, ref):
            assert torch.equal(a_, b_), f"{mode}, {label}: {name} differs from the recomputing passes by {(a_.float() - b_.float()).abs().max().item()}"


@pytest.mark.parametrize("rule", ["causal", "window"])
def test_a_built_in_mask_written_as_functions_is_bit_identical_to_the_built_in_kernels(rule):
    """A size-independent property at 4 x 2048 (the oracle takes minutes there): the causal mask written as a prefix function
    (key j < i + 1) and a local window written as one band (i - wl <= j < i + wr + 1) give the outputs and gradients of the causal /
    window kernels bit for bit -- same tiles, same MFMA order, the masked elements exact zeros either way (d = 256: two-wave-kind
    forward, exchange backward on both sides; the function side in the dense exchange layout, the causal side in the triangular one)."""
    from hstu import hstu_attn_varlen_func

    rng = np.random.default_rng(11)
    B, L, H, d = 4, 2048, 2, 256
    T = B * L
    mk = lambda lo, hi: torch.from_numpy(rng.uniform(lo, hi, (T, H, d)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(-1, 1), mk(-1, 1), mk(-1, 1), mk(0, 1)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=DEV)
    pos = (torch.arange(T, device=DEV) % L).to(torch.int32)
    if rule == "causal":
        f = (pos + 1).view(1, 1, T).contiguous()
        window = (-1, 0)
    else:
        wl, wr = 700, 90
        f = torch.stack([torch.zeros_like(pos), (pos - wl).clamp(min=0), pos + wr + 1]).view(1, 3, T).contiguous()
        window = (wl, wr)
    res = []
    for kw in (dict(window_size=(-1, -1), func=f), dict(window_size=window)):
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = hstu_attn_varlen_func(qq, kk, vv, cu, cu, None, None, L, L, L, None, None, alpha=1.0 / d ** 0.5, **kw)
        out.backward(dout)
        res.append((out.detach(), qq.grad, kk.grad, vv.grad))
    for name, a_, b_ in zip(("out", "dq", "dk", "dv"), *res):
        assert torch.equal(a_, b_), f"{rule}: {name} differs by {(a_.float() - b_.float()).abs().max().item()}"
    assert float(res[0][1].float().abs().max()) > 0


@pytest.mark.parametrize("d", [64, 256])
def test_mask_functions_with_contextual_rows_vs_oracle(d):
    """`func` together with num_contexts (refused while the functions were a bias): contextual rows see the whole history whatever
    the functions say -- the reference's context test `continue`s in front of every other mask (hstu_fwd.h:519-524).  (d = 256: the
    two-wave-kind forward and the exchange backward)"""
    from hstu import hstu_attn_varlen_func

    rng = np.random.default_rng(33)
    lengths = np.array([200, 3, 0, 129, 70])
    off, q, k, v, dout, f = _func_case(rng, lengths, 2, d, torch.bfloat16)
    B, N = lengths.size, int(lengths.max())
    f[:, 0] = rng.integers(0, 6, size=f[:, 0].shape)          # narrow first intervals: the exemption is what lets context rows see
    ctx = np.minimum(np.array([5, 2, 0, 9, 1]), lengths)
    tgt = np.minimum(np.array([7, 0, 0, 20, 3]), np.maximum(lengths - ctx, 0))
    cu = torch.from_numpy(off.astype(np.int32)).to(DEV)
    ti = lambda a: torch.from_numpy(a.astype(np.int32)).to(DEV)
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    alpha = 1.0 / 8
    out = hstu_attn_varlen_func(qq, kk, vv, cu, cu, None, None, N, N, N, ti(ctx), ti(tgt), target_group_size=2, window_size=(-1, 0),
                                alpha=alpha, func=torch.from_numpy(f).to(DEV))
    out.backward(dout)
    qn, kn, vn, dn = (t.float().cpu().numpy() for t in (q, k, v, dout))
    okw = dict(causal=True, num_targets=tgt, num_contextuals=ctx, target_group_size=2, func=f)
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, N, **okw)
    res = ho.hstu_attn_bwd(dn, qn, kn, vn, off, alpha, N, **okw)
    _assert_vs_oracle(out, (qq.grad, kk.grad, vv.grad), ref, res[0], res[1], res[2])
    # and the exemption matters in this case: without it the context rows' outputs differ
    ref_no = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, N, causal=True, num_targets=tgt, target_group_size=2, func=f)
    assert np.abs(ref_no - ref).max() > 1e-2


def test_mask_functions_allocate_no_mask_tensor():
    """2 x 4096 tokens: the forward under `func` allocates its output and nothing else (the dense statement of the same mask is a
    [2, 2, 4096, 4096] bias: 128 MB here, 1 GB at batch 32)"""
    from hstu import hstu_attn_varlen_func

    rng = np.random.default_rng(5)
    lengths = np.array([4096, 4096])
    off, q, k, v, dout, f = _func_case(rng, lengths, 2, 64, torch.bfloat16, per_head=False, slack=0)
    f[:, 3] = rng.integers(2000, 3000, size=f[:, 3].shape); f[:, 4] = f[:, 3] + rng.integers(0, 900, size=f[:, 3].shape)
    cu = torch.from_numpy(off.astype(np.int32)).to(DEV)
    func = torch.from_numpy(f).to(DEV)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        out = hstu_attn_varlen_func(q, k, v, cu, cu, None, None, 4096, 4096, 4096, None, None, window_size=(-1, -1), alpha=0.125,
                                    func=func)
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() - base < 2 * q.numel() * q.element_size()
    sel = np.r_[0:40, 4000:4096, 4096:4130, 8100:8192]          # a few hundred rows against the oracle (the whole batch takes minutes)
    qn, kn, vn = (t.float().cpu().numpy() for t in (q, k, v))
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, 0.125, 4096, causal=False, func=f)
    err = np.abs(out.float().cpu().numpy()[sel] - ref[sel]).max()
    assert err <= 6e-3 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("mode", ["causal", "ctx_targets", "noncausal", "window", "one_head"])
def test_rab_random_jagged_vs_oracle(d, mode):
    """bias with every mask family (contextual / target rows included: the op allows them with the causal mask), per-head
    and one shared head, over several key / query tiles, ragged ends, an empty and a 1-token sequence"""
    rng = np.random.default_rng(d + len(mode))
    lengths = np.array([300, 1, 0, 129, 64, 77])
    B, H, N = lengths.size, 2, int(lengths.max())
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    mk = lambda lo, hi, *shape: torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(0, 1, T, H, d)
    rab = mk(-2, 2, B, 1 if mode == "one_head" else H, N, N)
    targets = ctx = None
    grp, window = 1, (-1, 0)
    if mode == "ctx_targets":
        targets = np.minimum(rng.integers(0, 11, size=B), np.maximum(lengths - 1, 0))
        ctx = np.minimum(rng.integers(0, 5, size=B), np.maximum(lengths - 1 - targets, 0))
        grp = 2
    elif mode == "noncausal":
        window = (-1, -1)
    elif mode == "window":
        window = (40, 9)
    alpha = 1.0 / d ** 0.5
    out, grads, drab = _run_rab(q, k, v, rab, off, N, targets, ctx, grp, window, alpha, dout)
    qn, kn, vn, dn, rn = (t.float().cpu().numpy() for t in (q, k, v, dout, rab))
    kw = dict(causal=window == (-1, 0), num_targets=targets, num_contextuals=ctx, target_group_size=grp,
              local_window=window if mode == "window" else None, rab=rn)
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, N, **kw)
    dq, dk, dv, dr = ho.hstu_attn_bwd(dn, qn, kn, vn, off, alpha, N, **kw)
    _assert_vs_oracle(out, grads, ref, dq, dk, dv, ho.hstu_attn_magnitudes(dn, qn, kn, vn, off, alpha, N, **kw))
    _assert_drab(drab, dr)


def test_rab_without_drab_and_zero_bias():
    """has_drab=False: no gradient for rab; an all-zero bias reproduces the plain kernels bit for bit"""
    rng = np.random.default_rng(5)
    lengths = np.array([200, 64, 31])
    off = np.concatenate([[0], np.cumsum(lengths)])
    T, H, d, N = int(off[-1]), 2, 64, 200
    mk = lambda: torch.from_numpy(rng.uniform(-1, 1, (T, H, d)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    q, k, v, dout = mk(), mk(), mk(), mk()
    rab = torch.zeros(3, H, N, N, device=DEV, dtype=torch.bfloat16)
    out, grads, drab = _run_rab(q, k, v, rab, off, N, None, None, 1, (-1, 0), 0.125, dout, has_drab=False)
    assert drab is None
    o_p, g_p = _run(q, k, v, off, N, None, None, 1, True, 0.125, dout=dout)
    assert torch.equal(out, o_p)
    for a, b in zip(grads, g_p):
        assert torch.equal(a, b)


_FWD_VARIANTS = {
    "rows64": {"MI355_HSTU_FWD": "1"},                                   # hstu_fwd_q2_kernel at every length (default: from 1 025 rows)
    "rows64_pairs": {"MI355_HSTU_FWD": "2"},                             # ... with row blocks in pairs on every batch (default: dense ones)
    "rows32": {"MI355_HSTU_FWD": "3"},                                   # hstu_fwd_pc_kernel / hstu_fwd_pair_kernel at every length
    "rows32_pairs": {"MI355_HSTU_FWD": "4"},
    "one_stream": {"MI355_HSTU_FWD": "5"},                               # hstu_fwd_kernel, register-staged tiles
}


@pytest.mark.parametrize("variant", list(_FWD_VARIANTS))
def test_forward_kernel_variants(variant):
    """The d = 256 forward has four kernels (the one-kind LDS-DMA kernel of round 3 won on no shape and was removed in round 5).  Default since round 4: 8-wave workgroups of S waves and O waves (two waves per
    SIMD) -- hstu_fwd_q2_kernel (64 query rows per wave: two MFMAs per LDS fragment) from 1 025 rows per sequence,
    hstu_fwd_pc_kernel / hstu_fwd_pair_kernel (32 rows per wave) below; row blocks in (heavy, light) pairs on dense batches.
    MI355_HSTU_FWD (one test hook, values 1..5) forces each of them onto every shape; 5 is the one-kind register-staged kernel.
    The library reads the hook once, so every d = 256 test of this file (goldens, random jagged batches, contexts / targets, local windows, delta-q) is
    re-run in a child process with each kernel forced onto every shape."""
    import subprocess
    import sys

    if os.environ.get("MI355_HSTU_CHILD"):
        pytest.skip("already a variant run")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "256 and not mask_alone and not rab and not kernel_variants"],
                       env=dict(os.environ, MI355_HSTU_CHILD="1", **_FWD_VARIANTS[variant]), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout, r.stdout[-500:]


_LONG_CASES_SCRIPT = r"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(sys.argv[1], "recsys-examples_amd")); sys.path.insert(0, sys.argv[1])
from hstu import hstu_attn_varlen_func
dev = torch.device("cuda")
H, d = 2, 256
rng = np.random.default_rng(17)
def run(lengths, ctx, tgt, grp, window, klen_extra=None):
    lengths = np.asarray(lengths, np.int64)
    off = np.concatenate([[0], np.cumsum(lengths)])
    T, N = int(off[-1]), int(lengths.max())
    g = torch.Generator(device=dev); g.manual_seed(int(lengths.sum()))
    q, k, v = (torch.empty(T, H, d, device=dev).uniform_(-1, 1, generator=g).bfloat16() for _ in range(3))
    cu = torch.tensor(off, dtype=torch.int32, device=dev)
    ti = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=torch.int32, device=dev)
    out = hstu_attn_varlen_func(q, k, v, cu, cu, None, None, N, N, N, ti(ctx), ti(tgt), target_group_size=grp, window_size=window, alpha=1.0 / 16)
    torch.cuda.synchronize()
    return hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
L = [1025, 2300, 1, 1536, 64, 1100, 4096, 129]
print("causal", run(L, None, None, 1, (-1, 0)))
print("noncausal", run(L, None, None, 1, (-1, -1)))
print("ctx_tgt", run(L, [3, 70, 0, 128, 5, 0, 200, 1], [100, 64, 0, 7, 3, 500, 1000, 60], 1, (-1, 0)))
print("tgt_groups", run(L, None, [100, 64, 0, 7, 3, 500, 1000, 60], 4, (-1, 0)))
print("window", run(L, None, None, 1, (300, 0)))
print("window2", run(L, None, None, 1, (70, 33)))
"""


def test_forward_64_rows_per_wave_is_bit_identical_on_long_jagged_batches(tmp_path):
    """From 1 025 rows per sequence the forward runs hstu_fwd_q2_kernel (64 query rows per wave).  Same MFMA order per output
    element, same roundings as the 32-rows kernels: the outputs of a jagged batch with sequences of 1 .. 4 096 rows must agree
    bit for bit under every mask rule (causal, none, contexts + targets, target groups, local windows) -- both kernels run in
    child processes (the library reads MI355_HSTU_FWD once) and print a hash of the output bits."""
    import subprocess
    import sys

    script = tmp_path / "long_cases.py"
    script.write_text(_LONG_CASES_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for q2 in ("1", "0"):       # "1": the default rule (64-row waves on these lengths), "0": 32-row waves forced (hook 3)
        r = subprocess.run([sys.executable, str(script), root], env=dict(os.environ, MI355_HSTU_FWD="0" if q2 == "1" else "3"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[q2] = [ln for ln in r.stdout.splitlines() if ln and not ln.startswith("/opt")]
    assert len(outs["1"]) == 6 and outs["1"] == outs["0"], (outs["1"], outs["0"])


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


@pytest.mark.parametrize("d", [32, 128, 256])
@pytest.mark.parametrize("mode", ["causal", "ctx_targets", "noncausal", "window", "rab"])
def test_fp16_operands_vs_oracle(d, mode):
    """hstu_api.cpp:359-366 accepts fp16 as well as bf16: the fp16 build of the kernels (mi355_hstu_attn_*_f16:
    v_mfma_f32_32x32x16_f16, round-to-nearest-even packing) against the float64 oracle on the same fp16 inputs, element by
    element with fp16's ulp (2^-10) and rounding floor -- an eighth of the bf16 tolerance, so a bf16 instruction left in the
    fp16 path cannot pass."""
    rng = np.random.default_rng(7 * d + len(mode))
    lengths = np.array([300, 1, 0, 129, 64, 77])
    B, H, N = lengths.size, 2, int(lengths.max())
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    mk = lambda lo, hi, *shape: torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32)).to(DEV).to(torch.float16)
    q, k, v, dout = mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(-1, 1, T, H, d), mk(0, 1, T, H, d)
    targets = ctx = None
    grp, window, rab = 1, (-1, 0), None
    if mode == "ctx_targets":
        targets = np.minimum(rng.integers(0, 11, size=B), np.maximum(lengths - 1, 0))
        ctx = np.minimum(rng.integers(0, 5, size=B), np.maximum(lengths - 1 - targets, 0))
        grp = 2
    elif mode == "noncausal":
        window = (-1, -1)
    elif mode == "window":
        window = (40, 9)
    elif mode == "rab":
        rab = mk(-2, 2, B, H, N, N)
    alpha = 1.0 / d ** 0.5
    if rab is not None:
        out, grads, drab = _run_rab(q, k, v, rab, off, N, targets, ctx, grp, window, alpha, dout)
    else:
        out, grads = _run(q, k, v, off, N, targets, ctx, grp, window == (-1, 0), alpha, dout=dout, window=window)
    assert out.dtype == torch.float16 and all(g.dtype == torch.float16 for g in grads)
    qn, kn, vn, dn = (t.float().cpu().numpy() for t in (q, k, v, dout))
    kw = dict(causal=window == (-1, 0), num_targets=targets, num_contextuals=ctx, target_group_size=grp,
              local_window=window if mode == "window" else None, rab=None if rab is None else rab.float().cpu().numpy())
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, alpha, N, **kw)
    res = ho.hstu_attn_bwd(dn, qn, kn, vn, off, alpha, N, **kw)
    mags = ho.hstu_attn_magnitudes(dn, qn, kn, vn, off, alpha, N, **kw)
    for got, want, mag, kk in ((out, ref, mags[0], 2), (grads[0], res[0], mags[1], 4), (grads[1], res[1], mags[2], 4),
                               (grads[2], res[2], mags[3], 4)):
        _close_elementwise(got, want, mag, kk, bits=10)
    if rab is not None:
        gn = drab.detach().float().cpu().numpy()
        assert np.abs(gn - res[3]).max() <= 2e-3 * np.abs(res[3]).max() + 1e-6


@pytest.mark.parametrize("d", [64, 256])
def test_dense_long_batch_block_map_equals_the_jagged_one(d):
    """A dense batch of long sequences (every length == max_seqlen, >= 16 blocks per column, B * H a multiple of 8) takes the
    column-major block -> (sequence, head, rank) map, a batch with one more, short sequence the rotating one
    (seq_head_of_block): the shared sequences must come out bit-identical, forward and backward -- the map only decides
    where a block runs."""
    torch.manual_seed(d)
    B, H, L = 4, 2, 2048
    T = B * L
    q, k, v, dout = (torch.empty(T + 100, H, d, device=DEV).uniform_(-1, 1).bfloat16() for _ in range(4))
    off_d = np.arange(0, T + 1, L)
    off_j = np.concatenate([off_d, [T + 100]])
    alpha = 1.0 / d ** 0.5
    o_d, g_d = _run(q[:T], k[:T], v[:T], off_d, L, None, None, 1, True, alpha, dout=dout[:T], scaling=L)
    o_j, g_j = _run(q, k, v, off_j, L, None, None, 1, True, alpha, dout=dout, scaling=L)
    assert torch.equal(o_d, o_j[:T])
    for a_, b_ in zip(g_d, g_j):
        assert torch.equal(a_, b_[:T])
    # and against the float64 oracle on one (sequence, head) column
    qn, kn, vn = (t[L:2 * L, 1:2].float().cpu().numpy() for t in (q, k, v))
    ref = ho.hstu_attn_fwd(qn, kn, vn, np.array([0, L]), alpha, L, True, None, None, 1)
    got = o_d.detach()[L:2 * L, 1:2].float().cpu().numpy()
    assert np.abs(got - ref).max() <= 6e-3 * np.abs(ref).max() + 1e-6


def test_rejects_unsupported():
    from hstu import hstu_attn_varlen_func

    q = torch.zeros(4, 1, 48, device=DEV, dtype=torch.bfloat16)
    cu = torch.tensor([0, 4], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        hstu_attn_varlen_func(q, q, q, cu, cu, None, None, 4, 4, 4, None, None)  # head_dim 48
    q = torch.zeros(4, 1, 32, device=DEV, dtype=torch.bfloat16)
    nt = torch.tensor([1], dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError):   # targets need causal (the reference's own check and message, :238-245)
        hstu_attn_varlen_func(q, q, q, cu, cu, None, None, 4, 4, 4, None, nt, window_size=(-1, -1))
    with pytest.raises(RuntimeError):
        hstu_attn_varlen_func(q.float(), q.float(), q.float(), cu, cu, None, None, 4, 4, 4, None, None)


# ---------------------------------------------------------------------------------------- inference: delta-q / paged KV
def _paged_case(name):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hstu_paged_golden.npz"))
    return {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "_")}


@pytest.mark.parametrize("name", ["warm", "cold"])
def test_paged_kv_forward_matches_reference_golden(name):
    """hstu_attn_varlen_func(..., kv_cache=, page_offsets=, page_ids=, last_page_lens=) vs the reference's own
    _hstu_paged_kv_attention on the committed fixtures (bf16-representable inputs).  Same criterion as the other
    kernel-vs-reference checks of this file: max abs error within 6e-3 of the output scale (P is rounded to bf16
    before the second GEMM, exactly as in the reference kernel)."""
    from hstu import hstu_attn_varlen_func

    c = _paged_case(name)
    dev = "cuda"
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    q, k, v, cache = (t(c[x], torch.bfloat16) for x in ("q", "k", "v", "cache"))
    out = hstu_attn_varlen_func(q, k, v, t(c["q_off"]), t(c["k_off"]), None, None, 48, 48, float(c["scaling"]), None,
                                t(c["num_cand"]), target_group_size=1, window_size=(-1, 0), alpha=float(c["alpha"]),
                                kv_cache=cache, page_offsets=t(c["page_off"]), page_ids=t(c["page_ids"]),
                                last_page_lens=t(c["last_len"]))
    got = out.float().cpu().numpy()
    want = c["out"]
    err = np.abs(got - want).max()
    assert err <= 6e-3 * np.abs(want).max() + 1e-6, err


@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_delta_q_and_paged_random_vs_oracle(d):
    """random jagged batch: (a) delta-q without a cache (keys longer than queries), (b) the same keys served from a
    paged cache built with append_kvcache; both against the CPU oracle"""
    from hstu import append_kvcache, hstu_attn_varlen_func

    rng = np.random.default_rng(d)
    B, H, P = 5, 2, 16
    new_hist = rng.integers(1, 70, B)
    num_cand = rng.integers(1, 9, B)
    old = rng.integers(0, 120, B)
    old[0] = 0
    qlen = new_hist + num_cand
    cachelen = old + new_hist
    klen = cachelen + num_cand
    q_off = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int32)
    k_off = np.concatenate([[0], np.cumsum(klen)]).astype(np.int32)
    T, Tk = int(q_off[-1]), int(k_off[-1])
    mk = lambda n: torch.empty(n, H, d, device=DEV).uniform_(-1, 1).bfloat16()
    q, k_new, v_new = mk(T), mk(T), mk(T)            # q / k / v of the request: [new history | candidates] per sequence
    k_old, v_old = mk(int(old.sum())), mk(int(old.sum()))
    o_off = np.concatenate([[0], np.cumsum(old)])
    # full key sequences for the cache-less delta-q call and the oracle: old history, new history, candidates
    kf, vf = [], []
    for b in range(B):
        kf += [k_old[o_off[b]:o_off[b + 1]], k_new[q_off[b]:q_off[b + 1]]]
        vf += [v_old[o_off[b]:o_off[b + 1]], v_new[q_off[b]:q_off[b + 1]]]
    k_full, v_full = torch.cat(kf), torch.cat(vf)
    alpha, scaling = 1.0 / d ** 0.5, 200.0
    tgt = torch.from_numpy(num_cand.astype(np.int32)).to(DEV)
    cuq, cuk = torch.from_numpy(q_off).to(DEV), torch.from_numpy(k_off).to(DEV)
    ref = ho.hstu_attn_fwd_delta_q(q.float().cpu().numpy(), k_full.float().cpu().numpy(), v_full.float().cpu().numpy(), q_off,
                                   k_off, alpha, scaling, True, num_cand)
    out_a = hstu_attn_varlen_func(q, k_full, v_full, cuq, cuk, None, None, int(qlen.max()), int(klen.max()), scaling, None, tgt,
                                  window_size=(-1, 0), alpha=alpha)
    err = np.abs(out_a.float().cpu().numpy() - ref).max()
    assert err <= 6e-3 * np.abs(ref).max() + 1e-6, err
    # (b) paged: old history written page by page, new history appended with append_kvcache
    npages = int(((cachelen + P - 1) // P).sum())
    cache = torch.zeros(npages + 3, 2, P, H, d, dtype=torch.bfloat16, device=DEV)
    perm = rng.permutation(npages + 3)[:npages]
    page_ids, page_off, last = [], [0], []
    cursor = 0
    for b in range(B):
        n = int((cachelen[b] + P - 1) // P)
        pages = perm[cursor:cursor + n]
        cursor += n
        page_ids += pages.tolist()
        page_off.append(len(page_ids))
        last.append(int(cachelen[b] - (n - 1) * P))
        for j in range(int(old[b])):
            cache[pages[j // P], 0, j % P] = k_old[o_off[b] + j]
            cache[pages[j // P], 1, j % P] = v_old[o_off[b] + j]
    ti = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    batch_idx = np.repeat(np.arange(B), new_hist)
    positions = np.concatenate([old[b] + np.arange(new_hist[b]) for b in range(B)])
    cand_off = np.concatenate([[0], np.cumsum(num_cand)])
    nnz = ti([int(new_hist.sum())])
    append_kvcache(k_new, v_new, ti(batch_idx), ti(positions), ti(cand_off), nnz, 0, cache, ti(page_ids), ti(page_off), ti(last), 0)
    kc, vc, koff = ho.gather_paged_kv(k_new.float().cpu().numpy(), v_new.float().cpu().numpy(), q_off, num_cand,
                                      cache.float().cpu().numpy(), page_off, page_ids, last)
    np.testing.assert_array_equal(kc, k_full.float().cpu().numpy())   # append_kvcache put every token where the walk finds it
    np.testing.assert_array_equal(vc, v_full.float().cpu().numpy())
    out_b = hstu_attn_varlen_func(q, k_new, v_new, cuq, cuk, None, None, int(qlen.max()), int(klen.max()), scaling, None, tgt,
                                  window_size=(-1, 0), alpha=alpha, kv_cache=cache, page_offsets=ti(page_off),
                                  page_ids=ti(page_ids), last_page_lens=ti(last))
    assert torch.equal(out_a, out_b)   # same keys, same order of operations -> bit-identical


@pytest.mark.parametrize("d", [64, 256])
@pytest.mark.parametrize("window", [(20, 0), (7, 5), (0, 0), (-1, 9), (300, 0)])
def test_local_window_over_delta_q_and_paged_keys(d, window):
    """A local attention window composed with delta-q keys and with the paged cache (inference; the reference composes Is_local
    with the query offset Lk - Lq and Paged_KV, hstu_fwd.h:104-131,463-470,516-545): the window runs over ABSOLUTE positions.
    (a) contiguous keys longer than the queries against the oracle, (b) the same keys from a paged cache: bit-identical to (a)."""
    from hstu import append_kvcache, hstu_attn_varlen_func

    rng = np.random.default_rng(d + 7 * window[0] + window[1])
    B, H, P = 5, 2, 16
    new_hist = rng.integers(1, 90, B)
    num_cand = rng.integers(1, 9, B)
    old = rng.integers(0, 150, B)
    old[0] = 0
    qlen = new_hist + num_cand
    cachelen = old + new_hist
    klen = cachelen + num_cand
    q_off = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int32)
    k_off = np.concatenate([[0], np.cumsum(klen)]).astype(np.int32)
    T = int(q_off[-1])
    mk = lambda n: torch.empty(n, H, d, device=DEV).uniform_(-1, 1).bfloat16()
    q, k_new, v_new = mk(T), mk(T), mk(T)
    k_old, v_old = mk(int(old.sum())), mk(int(old.sum()))
    o_off = np.concatenate([[0], np.cumsum(old)])
    kf, vf = [], []
    for b in range(B):
        kf += [k_old[o_off[b]:o_off[b + 1]], k_new[q_off[b]:q_off[b + 1]]]
        vf += [v_old[o_off[b]:o_off[b + 1]], v_new[q_off[b]:q_off[b + 1]]]
    k_full, v_full = torch.cat(kf), torch.cat(vf)
    alpha, scaling = 1.0 / d ** 0.5, 100.0
    cuq, cuk = torch.from_numpy(q_off).to(DEV), torch.from_numpy(k_off).to(DEV)
    ref = ho.hstu_attn_fwd_delta_q(q.float().cpu().numpy(), k_full.float().cpu().numpy(), v_full.float().cpu().numpy(), q_off,
                                   k_off, alpha, scaling, window=window)
    out_a = hstu_attn_varlen_func(q, k_full, v_full, cuq, cuk, None, None, int(qlen.max()), int(klen.max()), scaling, None, None,
                                  window_size=window, alpha=alpha)
    err = np.abs(out_a.float().cpu().numpy() - ref).max()
    assert err <= 8e-3 * max(np.abs(ref).max(), 1e-3) + 1e-6, err     # (two bf16 roundings -- P and the output -- of values near the maximum)
    # (b) the same keys from the paged cache
    npages = int(((cachelen + P - 1) // P).sum())
    cache = torch.zeros(npages + 2, 2, P, H, d, dtype=torch.bfloat16, device=DEV)
    perm = rng.permutation(npages + 2)[:npages]
    page_ids, page_off, last = [], [0], []
    cursor = 0
    for b in range(B):
        n = int((cachelen[b] + P - 1) // P)
        pages = perm[cursor:cursor + n]
        cursor += n
        page_ids += pages.tolist()
        page_off.append(len(page_ids))
        last.append(int(cachelen[b] - (n - 1) * P))
        for j in range(int(old[b])):
            cache[pages[j // P], 0, j % P] = k_old[o_off[b] + j]
            cache[pages[j // P], 1, j % P] = v_old[o_off[b] + j]
    ti = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    batch_idx = np.repeat(np.arange(B), new_hist)
    positions = np.concatenate([old[b] + np.arange(new_hist[b]) for b in range(B)])
    cand_off = np.concatenate([[0], np.cumsum(num_cand)])
    append_kvcache(k_new, v_new, ti(batch_idx), ti(positions), ti(cand_off), ti([int(new_hist.sum())]), 0, cache, ti(page_ids),
                   ti(page_off), ti(last), 0)
    out_b = hstu_attn_varlen_func(q, k_new, v_new, cuq, cuk, None, None, int(qlen.max()), int(klen.max()), scaling, None, None,
                                  window_size=window, alpha=alpha, kv_cache=cache, page_offsets=ti(page_off),
                                  page_ids=ti(page_ids), last_page_lens=ti(last))
    assert torch.equal(out_a, out_b)


@pytest.mark.parametrize("d", [64, 256])
@pytest.mark.parametrize("mask", ["causal_targets", "window", "full"])
def test_rab_over_delta_q_and_paged_keys(d, mask):
    """A relative attention bias composed with delta-q keys and with the paged cache (inference; hstu_fwd.h Has_rab with the
    sequence offset and Paged_KV): rab[b][h][i][j] is indexed by ABSOLUTE positions.  (a) contiguous keys against the oracle,
    (b) the same keys from a paged cache: bit-identical to (a)."""
    from hstu import append_kvcache, hstu_attn_varlen_func

    rng = np.random.default_rng(d + len(mask))
    B, H, P = 4, 2, 16
    new_hist = rng.integers(1, 60, B)
    num_cand = rng.integers(1, 7, B)
    old = rng.integers(0, 100, B)
    old[1] = 0
    qlen, cachelen = new_hist + num_cand, old + new_hist
    klen = cachelen + num_cand
    q_off = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int32)
    k_off = np.concatenate([[0], np.cumsum(klen)]).astype(np.int32)
    T, Nk = int(q_off[-1]), int(klen.max())
    mk = lambda n: torch.empty(n, H, d, device=DEV).uniform_(-1, 1).bfloat16()
    q, k_new, v_new = mk(T), mk(T), mk(T)
    k_old, v_old = mk(int(old.sum())), mk(int(old.sum()))
    o_off = np.concatenate([[0], np.cumsum(old)])
    kf, vf = [], []
    for b in range(B):
        kf += [k_old[o_off[b]:o_off[b + 1]], k_new[q_off[b]:q_off[b + 1]]]
        vf += [v_old[o_off[b]:o_off[b + 1]], v_new[q_off[b]:q_off[b + 1]]]
    k_full, v_full = torch.cat(kf), torch.cat(vf)
    rab = (torch.randn(B, H if d == 64 else 1, Nk, Nk, device=DEV) * 2).bfloat16()
    alpha, scaling = 1.0 / d ** 0.5, 100.0
    cuq, cuk = torch.from_numpy(q_off).to(DEV), torch.from_numpy(k_off).to(DEV)
    if mask == "causal_targets":
        window, tgt_np, causal = (-1, 0), num_cand, True
    elif mask == "window":
        window, tgt_np, causal = (25, 3), None, False
    else:
        window, tgt_np, causal = (-1, -1), None, False
    tgt = None if tgt_np is None else torch.from_numpy(tgt_np.astype(np.int32)).to(DEV)
    ref = ho.hstu_attn_fwd_delta_q(q.float().cpu().numpy(), k_full.float().cpu().numpy(), v_full.float().cpu().numpy(), q_off, k_off,
                                   alpha, scaling, causal, tgt_np, window=window if mask == "window" else None,
                                   rab=rab.float().cpu().numpy())
    out_a = hstu_attn_varlen_func(q, k_full, v_full, cuq, cuk, None, None, int(qlen.max()), Nk, scaling, None, tgt,
                                  window_size=window, alpha=alpha, rab=rab)
    err = np.abs(out_a.float().cpu().numpy() - ref).max()
    assert err <= 6e-3 * max(np.abs(ref).max(), 1e-3) + 1e-6, err
    if mask == "full":
        return      # (the paged walk below serves candidates from k_new: same as causal_targets / window)
    npages = int(((cachelen + P - 1) // P).sum())
    cache = torch.zeros(npages + 1, 2, P, H, d, dtype=torch.bfloat16, device=DEV)
    perm = rng.permutation(npages + 1)[:npages]
    page_ids, page_off, last, cursor = [], [0], [], 0
    for b in range(B):
        n = int((cachelen[b] + P - 1) // P)
        pages = perm[cursor:cursor + n]
        cursor += n
        page_ids += pages.tolist()
        page_off.append(len(page_ids))
        last.append(int(cachelen[b] - (n - 1) * P))
        for j in range(int(old[b])):
            cache[pages[j // P], 0, j % P] = k_old[o_off[b] + j]
            cache[pages[j // P], 1, j % P] = v_old[o_off[b] + j]
    ti = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    append_kvcache(k_new, v_new, ti(np.repeat(np.arange(B), new_hist)), ti(np.concatenate([old[b] + np.arange(new_hist[b]) for b in range(B)])),
                   ti(np.concatenate([[0], np.cumsum(num_cand)])), ti([int(new_hist.sum())]), 0, cache, ti(page_ids), ti(page_off), ti(last), 0)
    out_b = hstu_attn_varlen_func(q, k_new, v_new, cuq, cuk, None, None, int(qlen.max()), Nk, scaling, None, tgt,
                                  window_size=window, alpha=alpha, rab=rab, kv_cache=cache, page_offsets=ti(page_off),
                                  page_ids=ti(page_ids), last_page_lens=ti(last))
    assert torch.equal(out_a, out_b)


def test_paged_kvcache_ops_registration():
    """torch.ops.paged_kvcache_ops.append_kvcache exists with the reference's schema and writes the cache"""
    import paged_kvcache_ops  # noqa: F401

    H, d, P = 2, 32, 8
    key = torch.randn(10, H, d, device=DEV).bfloat16()
    val = torch.randn(10, H, d, device=DEV).bfloat16()
    cache = torch.zeros(4, 2, P, H, d, device=DEV, dtype=torch.bfloat16)
    ti = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    # one sequence: 6 new-history tokens (positions 3..8 of its cache) + 4 candidates; pages [2, 0]
    out = torch.ops.paged_kvcache_ops.append_kvcache(key, val, ti([0] * 6), ti(list(range(3, 9))), ti([0, 4]), ti([6]), 0, cache,
                                                     ti([2, 0]), ti([0, 2]), ti([1]), 0)
    assert out.data_ptr() == cache.data_ptr()
    for i, pos in enumerate(range(3, 9)):
        page = [2, 0][pos // P]
        assert torch.equal(cache[page, 0, pos % P], key[i]) and torch.equal(cache[page, 1, pos % P], val[i])
    assert float(cache[1].abs().sum()) == 0 and float(cache[3].abs().sum()) == 0


def test_decode_step_is_hipgraph_capturable():
    """config 5: append the new history to the paged cache + paged attention, captured once in a HIP graph and
    replayed on new data (static buffers) -- the step has no host sync and no allocation-dependent control flow"""
    from hstu import append_kvcache, hstu_attn_varlen_func

    H, d, P, B = 2, 64, 16, 3
    rng = np.random.default_rng(1)
    new_hist, num_cand, old = np.array([5, 9, 3]), np.array([2, 1, 4]), np.array([20, 0, 33])
    qlen, cachelen = new_hist + num_cand, old + new_hist
    q_off = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int32)
    k_off = np.concatenate([[0], np.cumsum(cachelen + num_cand)]).astype(np.int32)
    T = int(q_off[-1])
    ti = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=DEV)
    npg = (cachelen + P - 1) // P
    page_ids = ti(rng.permutation(int(npg.sum())))
    page_off = ti(np.concatenate([[0], np.cumsum(npg)]))
    last = ti(cachelen - (npg - 1) * P)
    cuq, cuk, tgt = ti(q_off), ti(k_off), ti(num_cand)
    bidx = ti(np.repeat(np.arange(B), new_hist))
    pos = ti(np.concatenate([old[b] + np.arange(new_hist[b]) for b in range(B)]))
    cand_off = ti(np.concatenate([[0], np.cumsum(num_cand)]))
    nnz = ti([int(new_hist.sum())])
    cache = torch.randn(int(npg.sum()), 2, P, H, d, device=DEV).bfloat16()
    q, k, v = (torch.randn(T, H, d, device=DEV).bfloat16() for _ in range(3))
    out = torch.empty_like(q)

    def step():
        append_kvcache(k, v, bidx, pos, cand_off, nnz, 0, cache, page_ids, page_off, last, 0)
        out.copy_(hstu_attn_varlen_func(q, k, v, cuq, cuk, None, None, int(qlen.max()), 64, 100.0, None, tgt,
                                        window_size=(-1, 0), alpha=0.125, kv_cache=cache, page_offsets=page_off,
                                        page_ids=page_ids, last_page_lens=last))

    step()   # warm up (library load, attribute setting) outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            step()
    for _ in range(2):
        for t in (q, k, v):
            t.copy_(torch.randn_like(t.float()).bfloat16())
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        step()
        torch.cuda.synchronize()
        assert torch.equal(got, out)


def test_c3_full_size_properties():
    """BASELINE configs[2] attention shape (B=32, L=512, H=4, d=256, causal) at full size, through properties of the
    operator: causality (outputs do not see later tokens, bit-exact), batch independence, linearity in V, and the
    adjoint identity <dV, V> = <dO, O> that holds because O is linear in V."""
    from hstu import hstu_varlen_bwd, hstu_varlen_fwd

    B, L, H, d = 32, 512, 4, 256
    T = B * L
    torch.manual_seed(3)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=DEV)
    q, k, v, do = (torch.empty(T, H, d, device=DEV).uniform_(-1, 1).bfloat16() for _ in range(4))
    alpha = 1.0 / d ** 0.5
    fwd = lambda q_, k_, v_: hstu_varlen_fwd(q_, k_, v_, cu, L, L, None, None, 1, True, alpha)
    out = fwd(q, k, v)
    # causality: garbage in the last 100 tokens of every sequence leaves the first 412 outputs bit-identical
    k2, v2, q2 = k.clone(), v.clone(), q.clone()
    tail = (torch.arange(T, device=DEV) % L) >= L - 100
    for t in (k2, v2, q2):
        t[tail] = torch.randn_like(t[tail].float()).bfloat16() * 3
    out2 = fwd(q2, k2, v2)
    assert torch.equal(out[~tail], out2[~tail]) and not torch.equal(out[tail], out2[tail])
    # batch independence: rolling the sequences rolls the outputs
    roll = lambda t: t.view(B, L, H, d).roll(5, 0).reshape(T, H, d).contiguous()
    assert torch.equal(fwd(roll(q), roll(k), roll(v)), roll(out))
    # linearity in V (bf16 rounding of the two outputs bounds the error)
    va = torch.empty_like(v).uniform_(-1, 1)
    o_sum = fwd(q, k, (v.float() + va.float()).bfloat16()).float()
    o_parts = out.float() + fwd(q, k, va).float()
    assert (o_sum - o_parts).abs().max() <= 2.0 ** -6 * o_parts.abs().max() + 1e-6
    # adjoint identity of the backward, with dO = O so that the inner product is a well-conditioned positive number
    dq, dk, dv = hstu_varlen_bwd(out, q, k, v, cu, L, L, None, None, 1, True, alpha)
    lhs = (dv.double() * v.double()).sum()
    rhs = (out.double() * out.double()).sum()
    assert abs(float(lhs - rhs)) <= 5e-3 * abs(float(rhs)), (float(lhs), float(rhs))
    # the gradient of a causal operator is anti-causal: dK / dV of the last tokens only depend on the last queries
    do2 = do.clone()
    do2[~tail] = 0
    dq3, dk3, dv3 = hstu_varlen_bwd(do2, q, k, v, cu, L, L, None, None, 1, True, alpha)
    assert float(dq3[~tail].abs().max()) == 0.0


def test_c4_jagged_long_sequences_properties():
    """BASELINE config 4 attention shape: jagged lengths Zipf(1.2) clipped to [32, 4096], H=4, d=256, causal, with
    candidates (targets) at the end of every sequence.  Properties: a sequence computed ALONE gives bit-identical
    outputs and gradients to its slice of the batch (jagged addressing, dispatch order and block ranks do not leak
    between sequences); the adjoint identity <dV, V> = <O, O>."""
    from hstu import hstu_varlen_bwd, hstu_varlen_fwd

    rng = np.random.default_rng(4)
    B, H, d = 16, 4, 256
    lengths = np.clip(rng.zipf(1.2, B) + 31, 32, 4096).astype(np.int64)
    lengths[0], lengths[1] = 4096, 33
    ntgt = np.minimum(rng.integers(0, 9, B), lengths - 1).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    T, Lmax = int(offs[-1]), int(lengths.max())
    torch.manual_seed(8)
    q, k, v = (torch.empty(T, H, d, device=DEV).uniform_(-1, 1).bfloat16() for _ in range(3))
    cu = torch.from_numpy(offs).to(DEV)
    tg = torch.from_numpy(ntgt).to(DEV)
    alpha = 1.0 / d ** 0.5
    out = hstu_varlen_fwd(q, k, v, cu, Lmax, Lmax, None, tg, 1, True, alpha)
    dq, dk, dv = hstu_varlen_bwd(out, q, k, v, cu, Lmax, Lmax, None, tg, 1, True, alpha)
    for b in (0, 1, int(np.argsort(lengths)[B // 2])):
        lo, hi = int(offs[b]), int(offs[b + 1])
        cu1 = torch.tensor([0, hi - lo], dtype=torch.int32, device=DEV)
        qb, kb, vb = q[lo:hi].contiguous(), k[lo:hi].contiguous(), v[lo:hi].contiguous()
        o1 = hstu_varlen_fwd(qb, kb, vb, cu1, hi - lo, Lmax, None, tg[b:b + 1], 1, True, alpha)
        assert torch.equal(o1, out[lo:hi]), b
        g1 = hstu_varlen_bwd(out[lo:hi].contiguous(), qb, kb, vb, cu1, hi - lo, Lmax, None, tg[b:b + 1], 1, True, alpha)
        for got, ref in zip(g1, (dq, dk, dv)):
            assert torch.equal(got, ref[lo:hi]), b
    lhs = (dv.double() * v.double()).sum()
    rhs = (out.double() * out.double()).sum()
    assert abs(float(lhs - rhs)) <= 5e-3 * abs(float(rhs)), (float(lhs), float(rhs))


def test_c5_paged_decode_full_size_equals_contiguous_keys():
    """BASELINE config 5 at full size (page 32, 3968 cached + 128 new history + 256 candidates per sequence, d = 256):
    the paged-KV forward must be bit-identical to the delta-q forward over the same keys laid out contiguously (only the
    key addressing differs), after append_kvcache wrote the new history into randomly permuted pages."""
    from hstu import append_kvcache, hstu_attn_varlen_func

    B, H, d, P, old, new_hist, cand = 4, 4, 256, 32, 3968, 128, 256
    qlen, cachelen = new_hist + cand, old + new_hist
    klen = cachelen + cand
    npg = cachelen // P
    rng = np.random.default_rng(9)
    ti = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=DEV)
    page_ids = ti(rng.permutation(B * npg))
    page_off, last = ti(np.arange(B + 1) * npg), ti(np.full(B, P))
    cuq, cuk, tgt = ti(np.arange(B + 1) * qlen), ti(np.arange(B + 1) * klen), ti(np.full(B, cand))
    torch.manual_seed(2)
    cache = torch.zeros(B * npg, 2, P, H, d, device=DEV, dtype=torch.bfloat16)
    hist_k, hist_v = (torch.empty(B, old, H, d, device=DEV).uniform_(-1, 1).bfloat16() for _ in range(2))
    q, k, v = (torch.empty(B * qlen, H, d, device=DEV).uniform_(-1, 1).bfloat16() for _ in range(3))
    pid = page_ids.view(B, npg).long()
    for b in range(B):      # old history page by page (host-side fill of the fixture)
        pages = pid[b, : old // P]
        cache[pages, 0] = hist_k[b].view(old // P, P, H, d)
        cache[pages, 1] = hist_v[b].view(old // P, P, H, d)
    bidx = ti(np.repeat(np.arange(B), new_hist))
    pos = ti(np.tile(old + np.arange(new_hist), B))
    append_kvcache(k, v, bidx, pos, ti(np.arange(B + 1) * cand), ti([B * new_hist]), 0, cache, page_ids, page_off, last, 0)
    out_paged = hstu_attn_varlen_func(q, k, v, cuq, cuk, None, None, qlen, klen, float(klen), None, tgt, window_size=(-1, 0),
                                      alpha=1.0 / d ** 0.5, kv_cache=cache, page_offsets=page_off, page_ids=page_ids,
                                      last_page_lens=last)
    kq, vq = k.view(B, qlen, H, d), v.view(B, qlen, H, d)
    k_full = torch.cat([hist_k, kq], 1).reshape(B * klen, H, d).contiguous()
    v_full = torch.cat([hist_v, vq], 1).reshape(B * klen, H, d).contiguous()
    out_flat = hstu_attn_varlen_func(q, k_full, v_full, cuq, cuk, None, None, qlen, klen, float(klen), None, tgt,
                                     window_size=(-1, 0), alpha=1.0 / d ** 0.5)
    assert torch.equal(out_paged, out_flat)
    assert float(out_paged.float().abs().max()) > 0


def test_raw_fbgemm_ops_of_the_fused_layer_match_the_wrapper():
    """torch.ops.fbgemm.hstu_varlen_{fwd,bwd}_{80,90} with the positional order of examples/hstu/ops/fused_hstu_op.py
    :318-366,:682-750 give what hstu_attn_varlen_func + autograd give"""
    import hstu  # noqa: F401 (registers the ops)
    from hstu import hstu_attn_varlen_func

    torch.manual_seed(0)
    lens = [37, 0, 128, 5]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    T, H, D = int(cu[-1]), 2, 64
    q, k, v = (torch.randn(T, H, D, device="cuda", dtype=torch.bfloat16).requires_grad_() for _ in range(3))
    nt = torch.tensor([3, 0, 10, 1], dtype=torch.int32, device="cuda")
    nc = torch.tensor([2, 0, 4, 0], dtype=torch.int32, device="cuda")
    alpha, L = 1.0 / 8, 128
    ref = hstu_attn_varlen_func(q, k, v, cu, cu.clone(), None, None, L, L, L, nc, nt, 1, (-1, 0), alpha)
    dout = torch.randn_like(ref)
    gq, gk, gv = torch.autograd.grad(ref, (q, k, v), dout)
    with torch.no_grad():
        o80, rab = torch.ops.fbgemm.hstu_varlen_fwd_80(q, k, v, cu, cu, None, None, L, L, L, nc, nt, 1, -1, 0, alpha, None, None)
        o90, _ = torch.ops.fbgemm.hstu_varlen_fwd_90(q, k, v, cu, cu, None, None, L, L, L, nc, nt, 1, -1, 0, alpha, None, None, -1, 0)
        assert rab is None and torch.equal(o80, ref) and torch.equal(o90, ref)
        dq, dk, dv, drab = torch.ops.fbgemm.hstu_varlen_bwd_80(dout, q, k, v, cu, cu, None, None, L, L, L, None, None, None, nc, nt,
                                                               1, -1, 0, alpha, None, False, None, False)
        assert drab is None and torch.equal(dq, gq) and torch.equal(dk, gk) and torch.equal(dv, gv)
        bq = torch.empty_like(q)
        r = torch.ops.fbgemm.hstu_varlen_bwd_90(dout, None, q, None, k, None, v, cu, cu, None, None, L, L, L, bq, None, None, nc, nt,
                                                1, -1, 0, alpha, -1, None, False, None, None, None, None, None, None, None, None,
                                                None, None, None, None, 0, False)
        assert torch.equal(r[0], gq) and r[0].data_ptr() == bq.data_ptr() and torch.equal(r[1], gk) and torch.equal(r[2], gv)
        with pytest.raises(ValueError):   # contexts / targets with a window: undefined (hstu_api.cpp:163-164)
            torch.ops.fbgemm.hstu_varlen_fwd_80(q, k, v, cu, cu, None, None, L, L, L, nc, nt, 1, 16, 0, alpha, None, None)
    # bias through the raw ops == through the wrapper (the forward hands the bias back, the backward returns drab)
    rab = torch.randn(4, 1, L, L, device="cuda", dtype=torch.bfloat16).requires_grad_()
    ref = hstu_attn_varlen_func(q, k, v, cu, cu.clone(), None, None, L, L, L, nc, nt, 1, (-1, 0), alpha, rab=rab, has_drab=True)
    gq, gk, gv, gr = torch.autograd.grad(ref, (q, k, v, rab), dout)
    with torch.no_grad():
        o80, rab_back = torch.ops.fbgemm.hstu_varlen_fwd_80(q, k, v, cu, cu, None, None, L, L, L, nc, nt, 1, -1, 0, alpha, rab, None)
        assert torch.equal(o80, ref) and rab_back.data_ptr() == rab.data_ptr()
        dq, dk, dv, drab = torch.ops.fbgemm.hstu_varlen_bwd_80(dout, q, k, v, cu, cu, None, None, L, L, L, None, None, None, nc, nt,
                                                               1, -1, 0, alpha, rab, True, None, False)
        assert torch.equal(dq, gq) and torch.equal(dk, gk) and torch.equal(dv, gv) and torch.equal(drab, gr)
    # local window through the raw ops == through the wrapper
    ref = hstu_attn_varlen_func(q, k, v, cu, cu.clone(), None, None, L, L, L, None, None, 1, (16, 5), alpha)
    gq, gk, gv = torch.autograd.grad(ref, (q, k, v), dout)
    with torch.no_grad():
        o80, _ = torch.ops.fbgemm.hstu_varlen_fwd_80(q, k, v, cu, cu, None, None, L, L, L, None, None, 1, 16, 5, alpha, None, None)
        assert torch.equal(o80, ref)
        r = torch.ops.fbgemm.hstu_varlen_bwd_90(dout, None, q, None, k, None, v, cu, cu, None, None, L, L, L, None, None, None, None,
                                                None, 1, 16, 5, alpha, -1, None, False, None, None, None, None, None, None, None,
                                                None, None, None, None, None, 0, False)
        assert torch.equal(r[0], gq) and torch.equal(r[1], gk) and torch.equal(r[2], gv)
