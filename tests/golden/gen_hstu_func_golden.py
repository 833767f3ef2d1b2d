"""Generates tests/golden/hstu_func_golden.npz: HSTU attention under an ARBITRARY mask (`func`: per query token a set of key
intervals, hstu_api.cpp:170-180, applied in hstu_fwd.h:493-556) as the REFERENCE's own test states it -- `construct_mask` with
`func` (/root/reference/corelib/hstu/test.py:101-140: row i of sequence b sees the columns [0, func[b, 0, 0, i]) and
[func[b, 0, 2p - 1, i], func[b, 0, 2p, i])), `_hstu_attention_maybe_from_cache` (:584-664) and the pad helpers pulled out of that
file's AST and run on CPU in fp32; gradients by autograd through the extracted function.  The mask functions are the ones the
reference test builds (:406-505): three random interval bounds per token, the causal emulation (one bound, token + 1), the
local-window emulation (2 left, 12 right).  Stored in the KERNEL's layout (:515-533 of the test): [heads_func, n_func, total_q].
Run in the build container only:

    python tests/golden/gen_hstu_func_golden.py
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
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hstu_func_golden.npz")
WANT = {"pad_input", "unpad_input", "construct_mask", "_hstu_attention_maybe_from_cache"}
ns = {"torch": torch, "F": F, "rearrange": rearrange, "Optional": Optional, "Tuple": Tuple, "math": math, "debug": False}
for node in ast.parse(open(REF).read()).body:
    if isinstance(node, ast.FunctionDef) and node.name in WANT:
        exec(compile(ast.Module([node], []), REF, "exec"), ns)

CASES = [
    # name, lengths, H, d, kind
    ("random3", [70, 9, 33, 1], 2, 32, "random3"),
    ("random5", [64, 40, 17], 2, 64, "random5"),
    ("causal_emulation", [70, 9, 33], 2, 32, "causal"),
    ("local_emulation", [90, 31, 64], 1, 128, "local"),
    ("random3_d256", [66, 35], 1, 256, "random3"),
]


def make_func(kind, B, N, gen):
    """the constructions of test.py:406-470, batch x 1 head x n_func x max_seqlen"""
    if kind.startswith("random"):
        n_func = int(kind[6:])
        split = N // n_func
        f = torch.empty(B, 1, n_func, N, dtype=torch.int32)
        for i in range(n_func):
            f[:, :, i, :] = torch.randint(i * split, max(int((i + 0.3) * split), i * split + 1), (B, 1, N), generator=gen)
        return f
    tok = torch.arange(N, dtype=torch.int32)
    if kind == "causal":
        return (tok + 1).view(1, 1, 1, N).expand(B, 1, 1, N).contiguous()
    f = torch.zeros(B, 1, 3, N, dtype=torch.int32)
    f[:, :, 1, :] = torch.clamp(tok - 2, min=0)
    f[:, :, 2, :] = torch.clamp(tok + 12 + 1, max=N)
    return f


def run(case):
    name, lengths, H, d, kind = case
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2**31))
    off = torch.tensor([0] + list(np.cumsum(lengths)), dtype=torch.int32)
    T, N, B = int(off[-1]), max(lengths), len(lengths)
    mk = lambda *shape, lo=-1.0, hi=1.0: torch.empty(*shape).uniform_(lo, hi, generator=gen).bfloat16().float()
    q, k, v = mk(T, H, d), mk(T, H, d), mk(T, H, d)
    dout = mk(T, H, d, lo=0.0)
    func = make_func(kind, B, N, gen)
    mask = ns["construct_mask"](batch_func=B, seqlen_c=0, seqlen=N, seqlen_t=0, target_group_size=1, window_size=(-1, -1),
                                func=func, cu_seqlens_q=off, cu_seqlens_k=off, num_contexts=None,
                                device=torch.device("cpu")).to(torch.float32)
    qq, kk, vv = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out = ns["_hstu_attention_maybe_from_cache"](
        num_heads=H, attention_dim=d, linear_dim=d, seqlen_q=N, seqlen_k=N, q=qq.view(T, H * d), k=kk.view(T, H * d),
        v=vv.view(T, H * d), q_offsets=off, k_offsets=off, rab=None, invalid_attn_mask=mask, alpha=1.0 / d**0.5, upcast=True,
        is_delta_q=False)
    out.backward(dout)
    # the kernel's layout (test.py:515-533): [heads_func, n_func, total_q (+ padding)], token-major like q
    var = torch.zeros(1, func.shape[2], T + 256, dtype=torch.int32)
    for b, L in enumerate(lengths):
        var[0, :, int(off[b]): int(off[b + 1])] = func[b, 0, :, :L]
    return dict(q=q, k=k, v=v, dout=dout, func=var, off=off.to(torch.int64)), N, out.detach(), qq.grad, kk.grad, vv.grad


def main():
    blob, names = {}, []
    for case in CASES:
        name, lengths, H, d, kind = case
        inp, N, out, dq, dk, dv = run(case)
        for kname, t in inp.items():
            blob[f"{name}/{kname}"] = t.numpy()
        blob[f"{name}/meta"] = np.array([H, d, N], np.int32)
        for kname, t in (("out", out), ("dq", dq), ("dk", dk), ("dv", dv)):
            blob[f"{name}/{kname}"] = t.numpy().astype(np.float32)
        names.append(name)
    blob["cases"] = np.array(names)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB", names)


if __name__ == "__main__":
    main()
