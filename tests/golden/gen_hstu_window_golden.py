"""Generates tests/golden/hstu_window_golden.npz: local (sliding window) HSTU attention as the REFERENCE's own test states
it.  `construct_mask` (local branch, /root/reference/corelib/hstu/test.py:185-192), `pad_input` / `unpad_input` (:45-90)
and `_hstu_attention_maybe_from_cache` (:584-664) are pulled out of that file's AST and run on CPU in fp32 (the file
itself imports the CUDA `hstu_attn` package); gradients come from autograd through the extracted function.  The window
pairs are the reference's own parametrisation (:688-689: (111, 11), (111, 222)) plus one-sided ones written the way the
API spells "unbounded" (-1).  A side of exactly 0 is left out on purpose: the test's mask builder treats 0 as unbounded
(:187-188 `> 0 else seqlen`) while the op itself keeps 0 as a zero-width side (hstu_api.cpp:154-155 `< 0`), and the op
is what we follow.
Run in the build container only:

    python tests/golden/gen_hstu_window_golden.py
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
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hstu_window_golden.npz")
WANT = {"pad_input", "unpad_input", "construct_mask", "_hstu_attention_maybe_from_cache"}
ns = {"torch": torch, "F": F, "rearrange": rearrange, "Optional": Optional, "Tuple": Tuple, "math": math, "debug": False}
for node in ast.parse(open(REF).read()).body:
    if isinstance(node, ast.FunctionDef) and node.name in WANT:
        exec(compile(ast.Module([node], []), REF, "exec"), ns)

CASES = [
    # name, lengths, H, d, (left, right)
    ("w111_11", [256, 130, 17], 1, 32, (111, 11)),
    ("w111_222", [256, 130, 17], 1, 32, (111, 222)),
    ("w7_3_d64", [40, 9, 75, 1], 2, 64, (7, 3)),
    ("left_only_unbounded_right", [33, 70, 5], 1, 32, (9, -1)),
    ("right_only_unbounded_left", [33, 70, 5], 1, 32, (-1, 4)),
    ("w40_20_d128", [100, 64, 37], 1, 128, (40, 20)),
    ("w33_65_d256", [100, 31, 50], 1, 256, (33, 65)),
]


def run(case):
    name, lengths, H, d, win = case
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2**31))
    off = torch.tensor([0] + list(np.cumsum(lengths)), dtype=torch.int32)
    T, N = int(off[-1]), max(lengths)
    mk = lambda: torch.empty(T, H, d).uniform_(-1.0, 1.0, generator=gen).bfloat16().float()
    q, k, v = mk(), mk(), mk()
    dout = torch.empty(T, H, d).uniform_(0.0, 1.0, generator=gen).bfloat16().float()
    mask = ns["construct_mask"](batch_func=1, seqlen_c=0, seqlen=N, seqlen_t=0, target_group_size=1, window_size=win,
                                func=None, cu_seqlens_q=off, cu_seqlens_k=off, num_contexts=None,
                                device=torch.device("cpu"))
    qq, kk, vv = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out = ns["_hstu_attention_maybe_from_cache"](
        num_heads=H, attention_dim=d, linear_dim=d, seqlen_q=N, seqlen_k=N, q=qq.view(T, H * d), k=kk.view(T, H * d),
        v=vv.view(T, H * d), q_offsets=off, k_offsets=off, rab=None, invalid_attn_mask=mask.to(torch.float32),
        alpha=1.0 / d**0.5, upcast=True, is_delta_q=False)
    out.backward(dout)
    return dict(q=q, k=k, v=v, dout=dout, off=off.to(torch.int64)), N, mask, out.detach(), qq.grad, kk.grad, vv.grad


def main():
    blob, names = {}, []
    for case in CASES:
        name, lengths, H, d, win = case
        inp, N, mask, out, dq, dk, dv = run(case)
        for kname, t in inp.items():
            blob[f"{name}/{kname}"] = t.numpy()
        blob[f"{name}/meta"] = np.array([H, d, win[0], win[1], N], np.int32)
        blob[f"{name}/mask"] = np.packbits(mask.numpy().astype(np.uint8), axis=None)
        for kname, t in (("out", out), ("dq", dq), ("dk", dk), ("dv", dv)):
            blob[f"{name}/{kname}"] = t.numpy().astype(np.float32)
        names.append(name)
    blob["cases"] = np.array(names)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB", names)


if __name__ == "__main__":
    main()
