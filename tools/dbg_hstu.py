import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hstu import hstu_varlen_fwd, hstu_varlen_bwd
from oracle import hstu_oracle as ho
for d in (64, 128, 256):
  for L in (300, 64, 33, 128, 129):
    rng = np.random.default_rng(d)
    B, H = 2, 1
    lengths = np.array([L, 1])
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    mk = lambda lo, hi: torch.empty(T, H, d, device="cuda").uniform_(lo, hi).bfloat16()
    q, k, v, dout = mk(-1, 1), mk(-1, 1), mk(-1, 1), mk(0, 1)
    cu = torch.from_numpy(off.astype(np.int32)).cuda()
    alpha = 1.0 / d ** 0.5
    dq, dk, dv = hstu_varlen_bwd(dout, q, k, v, cu, L, L, None, None, 1, True, alpha)
    qn, kn, vn, dn = (t.float().cpu().numpy() for t in (q, k, v, dout))
    rq, rk, rv = ho.hstu_attn_bwd(dn, qn, kn, vn, off, alpha, L, True, None, None, 1)
    for name, got, want in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        gn = got.float().cpu().numpy()
        err = np.abs(gn - want)
        bad = np.argwhere(err > 0.02 * np.abs(want).max())
        print(d, L, name, "maxerr %.4f scale %.3f" % (err.max(), np.abs(want).max()), "bad rows:", sorted(set(bad[:, 0].tolist()))[:12], "n", len(bad))
