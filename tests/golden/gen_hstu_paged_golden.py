"""Generates tests/golden/hstu_paged_golden.npz with the REFERENCE's own statement of paged-KV HSTU attention:
`_hstu_paged_kv_attention` / `_hstu_attention_maybe_from_cache` (and their pad helpers) of
/root/reference/examples/hstu/test/test_paged_hstu_attn_kernel.py:36-256 are pulled out of that file's AST and run
on CPU (the file itself cannot be imported: it needs the CUDA `hstu` package at import time).  The inputs, the page
tables and the mask are built here the way the reference test builds them (:283-343 cache set-up, :455-484 mask).
Run in the build container only:

    python tests/golden/gen_hstu_paged_golden.py
"""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F  # noqa: F401  (used by the extracted functions)
from einops import rearrange  # noqa: F401
from typing import Optional  # noqa: F401

REF = "/root/reference/examples/hstu/test/test_paged_hstu_attn_kernel.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hstu_paged_golden.npz")
WANT = {"pad_input", "unpad_input", "pad_input_delta_q", "unpad_input_delta_q", "_hstu_attention_maybe_from_cache",
        "_hstu_paged_kv_attention"}
ns = {"torch": torch, "F": F, "rearrange": rearrange, "Optional": Optional}
for node in ast.parse(open(REF).read()).body:
    if isinstance(node, ast.FunctionDef) and node.name in WANT:
        exec(compile(ast.Module([node], []), REF, "exec"), ns)

torch.manual_seed(7)
rng = np.random.default_rng(7)
B, H, D, P, NPAGES = 4, 2, 32, 4, 64
MAXLEN = 48
bf = lambda t: t.to(torch.bfloat16).float()   # bf16-representable fp32 values

cases = {}
for name, cached_extra in (("warm", [5, 0, 9, 13]), ("cold", [0, 0, 0, 0])):
    new_hist = torch.tensor(rng.integers(1, 9, B), dtype=torch.int32)
    num_cand = torch.tensor(rng.integers(1, 5, B), dtype=torch.int32)
    old_hist = torch.tensor(cached_extra, dtype=torch.int32)     # tokens cached by earlier requests
    cachelen = old_hist + new_hist                               # the new history is appended before attention
    qlen = new_hist + num_cand
    q_off = torch.zeros(B + 1, dtype=torch.int32); q_off[1:] = torch.cumsum(qlen, 0)
    k_off = torch.zeros(B + 1, dtype=torch.int32); k_off[1:] = torch.cumsum(cachelen + num_cand, 0)
    T = int(q_off[-1])
    q, k, v = (bf(torch.randn(T, H, D)) for _ in range(3))
    cache = torch.zeros(NPAGES, 2, P, H, D)
    page_ids, page_off, last_len = [], [0], []
    free = list(rng.permutation(NPAGES))
    old_k = [bf(torch.randn(int(o), H, D)) for o in old_hist]
    old_v = [bf(torch.randn(int(o), H, D)) for o in old_hist]
    for b in range(B):
        L = int(cachelen[b]); nh = int(new_hist[b])
        kk = torch.cat([old_k[b], k[int(q_off[b]):int(q_off[b]) + nh]])
        vv = torch.cat([old_v[b], v[int(q_off[b]):int(q_off[b]) + nh]])
        npg = (L + P - 1) // P
        pages = [int(free.pop()) for _ in range(npg)]
        for pi, pg in enumerate(pages):
            n = min(P, L - pi * P)
            cache[pg, 0, :n] = kk[pi * P: pi * P + n]
            cache[pg, 1, :n] = vv[pi * P: pi * P + n]
        page_ids += pages
        page_off.append(len(page_ids))
        last_len.append(L - (npg - 1) * P)
    page_off_t = torch.tensor(page_off, dtype=torch.int32)
    page_ids_t = torch.tensor(page_ids, dtype=torch.int32)
    last_len_t = torch.tensor(last_len, dtype=torch.int32)
    # mask exactly as test_paged_hstu_attn_kernel.py:455-484
    mask = torch.zeros(B, H, MAXLEN, MAXLEN)
    for b in range(B):
        ql, cl, nc = int(qlen[b]), int(cachelen[b]), int(num_cand[b])
        seq_mask = torch.cat([torch.tril(torch.ones((ql, cl), dtype=torch.int32), diagonal=cl + nc - ql),
                              torch.cat([torch.zeros((ql - nc, nc), dtype=torch.int32), torch.eye(nc, dtype=torch.int32)], dim=0)],
                             dim=1)
        mask[b, :, :ql, :cl + nc] = seq_mask.float()
    alpha, scaling = 1.0 / D ** 0.5, 40
    out = ns["_hstu_paged_kv_attention"](
        num_heads=H, attention_dim=D, linear_dim=D, seqlen_q=MAXLEN, seqlen_k=MAXLEN, scaling_seqlen=scaling, q=q, k=k, v=v,
        q_offsets=q_off, k_offsets=k_off, num_targets=num_cand, invalid_attn_mask=mask, alpha=alpha, upcast=True,
        kv_cache=cache, page_offsets=page_off_t, page_ids=page_ids_t, last_page_lens=last_len_t).view(-1, H, D)
    for key, val in dict(q=q, k=k, v=v, cache=cache, q_off=q_off, k_off=k_off, num_cand=num_cand, page_off=page_off_t,
                         page_ids=page_ids_t, last_len=last_len_t, out=out, new_hist=new_hist).items():
        cases[f"{name}_{key}"] = val.numpy()
    cases[f"{name}_alpha"] = np.float32(alpha)
    cases[f"{name}_scaling"] = np.float32(scaling)
np.savez_compressed(OUT, **cases)
print("wrote", OUT, {k: v.shape for k, v in cases.items() if k.endswith("_out")})
