"""Pins the HSTU attention oracle (oracle/hstu_oracle.py) against golden vectors produced by the
reference's own PyTorch statement `pytorch_hstu_mha` (tests/golden/gen_hstu_golden.py): forward and
backward, all mask variants (causal, targets, contextuals, target groups, non causal)."""
import os

import numpy as np
import pytest

from oracle import hstu_oracle as ho

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "hstu_golden.npz"))
CASES = [str(c) for c in G["cases"]]


def load(name):
    g = lambda k: G[f"{name}/{k}"]
    H, d, causal, grp, N = [int(x) for x in g("meta")]
    t = g("targets"); c = g("ctx")
    return dict(q=g("q"), k=g("k"), v=g("v"), dout=g("dout"), off=g("off"), H=H, d=d, causal=bool(causal), grp=grp, N=N,
                targets=None if t[0] < 0 else t, ctx=None if c[0] < 0 else c,
                out=g("out"), dq=g("dq"), dk=g("dk"), dv=g("dv"),
                out_bf16=g("out_bf16"), dq_bf16=g("dq_bf16"), dk_bf16=g("dk_bf16"), dv_bf16=g("dv_bf16"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_pytorch_statement(name):
    c = load(name)
    alpha = 1.0 / c["d"] ** 0.5
    out = ho.hstu_attn_fwd(c["q"], c["k"], c["v"], c["off"], alpha, c["N"], c["causal"], c["targets"], c["ctx"], c["grp"])
    np.testing.assert_allclose(out, c["out"], rtol=2e-4, atol=2e-6)
    dq, dk, dv = ho.hstu_attn_bwd(c["dout"], c["q"], c["k"], c["v"], c["off"], alpha, c["N"], c["causal"], c["targets"],
                                  c["ctx"], c["grp"])
    np.testing.assert_allclose(dv, c["dv"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(dq, c["dq"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(dk, c["dk"], rtol=2e-4, atol=2e-6)
    # the reference's own acceptance rule (examples/commons/utils/hstu_assert_close.py:20-56) holds trivially
    assert np.abs(out - c["out"]).max() <= 2 * np.abs(c["out_bf16"] - c["out"]).max()


def test_mask_semantics():
    m = ho.valid_mask(8, True, 3, 2, 1)  # c=2 contexts, t=3 targets, h=5
    assert m[0, 1] and m[1, 0] and m[0, 4] and not m[0, 5]      # context rows see all history, no targets
    assert m[3, 2] and not m[2, 3]                               # causal inside history
    assert m[5, 4] and m[5, 5] and not m[6, 5] and not m[5, 6]   # group size 1: a target sees itself only
    assert not m[4, 5]                                           # history never sees targets
    m2 = ho.valid_mask(8, True, 4, 0, 2)
    assert m2[5, 4] and not m2[6, 5] and m2[7, 6]                # groups {4,5} {6,7}
    m3 = ho.valid_mask(4, False, None, None)
    assert m3.all()
