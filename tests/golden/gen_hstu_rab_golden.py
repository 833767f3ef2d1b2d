"""Generates tests/golden/hstu_rab_golden.npz: HSTU attention with a relative attention bias (`rab`, and its gradient `drab`)
as the REFERENCE's own test states it -- `_hstu_attention_maybe_from_cache` (`qk_attn + rab`, then `* alpha`, SiLU, `/ seqlen`,
mask; /root/reference/corelib/hstu/test.py:584-664), `construct_mask` (:101-193) and the pad helpers (:45-90) pulled out of
that file's AST and run on CPU in fp32; drab by autograd through the extracted function, as the reference test takes it
(:1111-1126).  Bias heads as in the reference's parametrisation (:708-716: heads_rab = heads, or 1 shared head).
Run in the build container only:

    python tests/golden/gen_hstu_rab_golden.py
"""
import ast
import math
import os
import zlib

import numpy as np
import torch
import torch.nn.functional as F
from einops import rearrange
from typing import Optional, Tuple

REF = "/root/reference/corelib/hstu/test.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hstu_rab_golden.npz")
WANT = {"pad_input", "unpad_input", "construct_mask", "_hstu_attention_maybe_from_cache"}
ns = {"torch": torch, "F": F, "rearrange": rearrange, "Optional": Optional, "Tuple": Tuple, "math": math, "debug": False}
for node in ast.parse(open(REF).read()).body:
    if isinstance(node, ast.FunctionDef) and node.name in WANT:
        exec(compile(ast.Module([node], []), REF, "exec"), ns)

CASES = [
    # name, lengths, H, heads_rab, d, window
    ("causal_rab", [70, 9, 33, 1], 2, 2, 32, (-1, 0)),
    ("causal_rab_one_head", [70, 9, 33], 2, 1, 32, (-1, 0)),
    ("full_rab", [40, 17, 5], 2, 2, 64, (-1, -1)),
    ("window_rab_one_head", [90, 31, 64], 2, 1, 32, (20, 7)),
    ("causal_rab_d128", [70, 40], 1, 1, 128, (-1, 0)),
    ("causal_rab_d256", [66, 35], 1, 1, 256, (-1, 0)),
]


def run(case):
    name, lengths, H, HR, d, win = case
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2**31))
    off = torch.tensor([0] + list(np.cumsum(lengths)), dtype=torch.int32)
    T, N, B = int(off[-1]), max(lengths), len(lengths)
    mk = lambda *shape, lo=-1.0, hi=1.0: torch.empty(*shape).uniform_(lo, hi, generator=gen).bfloat16().float()
    q, k, v = mk(T, H, d), mk(T, H, d), mk(T, H, d)
    dout = mk(T, H, d, lo=0.0)
    rab = mk(B, HR, N, N, lo=-2.0, hi=2.0)
    mask = None
    if win != (-1, -1):
        mask = ns["construct_mask"](batch_func=1, seqlen_c=0, seqlen=N, seqlen_t=0, target_group_size=1, window_size=win,
                                    func=None, cu_seqlens_q=off, cu_seqlens_k=off, num_contexts=None,
                                    device=torch.device("cpu")).to(torch.float32)
    qq, kk, vv, rr = [t.clone().requires_grad_(True) for t in (q, k, v, rab)]
    out = ns["_hstu_attention_maybe_from_cache"](
        num_heads=H, attention_dim=d, linear_dim=d, seqlen_q=N, seqlen_k=N, q=qq.view(T, H * d), k=kk.view(T, H * d),
        v=vv.view(T, H * d), q_offsets=off, k_offsets=off, rab=rr, invalid_attn_mask=mask, alpha=1.0 / d**0.5, upcast=True,
        is_delta_q=False)
    out.backward(dout)
    # the reference statement also differentiates through the zero-padded rows of a short sequence: rows / columns past a
    # sequence's length see zero-padded q / k, their d rab is the op's "no write" region (hstu_api.cpp:662-666) -> zero it
    drab = rr.grad.clone()
    for b, L in enumerate(lengths):
        drab[b, :, L:, :] = 0
        drab[b, :, :, L:] = 0
    return dict(q=q, k=k, v=v, dout=dout, rab=rab, off=off.to(torch.int64)), N, out.detach(), qq.grad, kk.grad, vv.grad, drab


def main():
    blob, names = {}, []
    for case in CASES:
        name, lengths, H, HR, d, win = case
        inp, N, out, dq, dk, dv, drab = run(case)
        for kname, t in inp.items():
            blob[f"{name}/{kname}"] = t.numpy()
        blob[f"{name}/meta"] = np.array([H, HR, d, win[0], win[1], N], np.int32)
        for kname, t in (("out", out), ("dq", dq), ("dk", dk), ("dv", dv), ("drab", drab)):
            blob[f"{name}/{kname}"] = t.numpy().astype(np.float32)
        names.append(name)
    blob["cases"] = np.array(names)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB", names)


if __name__ == "__main__":
    main()
