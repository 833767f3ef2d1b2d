"""Generates tests/golden/hstu_golden.npz by IMPORTING the reference's own PyTorch statement of HSTU
attention (`pytorch_hstu_mha`, /root/reference/examples/hstu/ops/pt_ops/pt_hstu_attention.py:149-196)
on CPU.  The reference module needs `fbgemm_gpu` only for two jagged<->dense helpers, which are defined
here with torch.library exactly as their documented semantics (pad with 0 / gather valid rows).
Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/gen_hstu_golden.py

Fixtures: inputs (q, k, v bf16-representable fp32, offsets, num_targets, num_contextuals, dout) and the
reference outputs in fp32 arithmetic (out, dq, dk, dv) and bf16 arithmetic (out_bf16, d*_bf16).
"""
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hstu_golden.npz")

# --- minimal stand-ins for the two fbgemm ops the reference file calls -------------------------
sys.modules.setdefault("fbgemm_gpu", types.ModuleType("fbgemm_gpu"))
lib = torch.library.Library("fbgemm", "DEF")
lib.define("jagged_to_padded_dense(Tensor values, Tensor[] offsets, int[] max_lengths, float padding_value=0.0) -> Tensor")
lib.define("dense_to_jagged(Tensor dense, Tensor[] offsets, int? total_L=None) -> Tensor[]")


def _j2d(values, offsets, max_lengths, padding_value=0.0):
    off = offsets[0]
    B, N = off.numel() - 1, max_lengths[0]
    idx = torch.arange(N).view(1, N) + off[:-1].view(B, 1)
    valid = torch.arange(N).view(1, N) < (off[1:] - off[:-1]).view(B, 1)
    idx = torch.where(valid, idx, torch.zeros_like(idx))
    out = values[idx.flatten()].view(B, N, -1)
    return torch.where(valid.unsqueeze(-1), out, torch.full_like(out, padding_value))


def _d2j(dense, offsets, total_L=None):
    off = offsets[0]
    B, N = dense.shape[0], dense.shape[1]
    valid = torch.arange(N).view(1, N) < (off[1:] - off[:-1]).view(B, 1)
    return [dense[valid]]


lib.impl("jagged_to_padded_dense", _j2d, "CompositeImplicitAutograd")
lib.impl("dense_to_jagged", _d2j, "CompositeImplicitAutograd")

sys.path.insert(0, "/root/reference/examples/hstu")
from ops.pt_ops.pt_hstu_attention import pytorch_hstu_mha  # noqa: E402

CASES = [
    # name, lengths, targets, contextuals, H, d, causal, target_group_size
    ("causal_plain", [5, 17, 1, 33], None, None, 2, 32, True, 1),
    ("causal_targets", [9, 20, 6, 40], [2, 5, 0, 7], None, 2, 32, True, 1),
    ("causal_ctx_targets", [12, 30, 7, 64], [3, 4, 2, 10], [2, 0, 4, 3], 2, 64, True, 1),
    ("causal_ctx_targets_group2", [12, 31, 8, 50], [4, 6, 2, 9], [1, 3, 0, 2], 1, 32, True, 2),
    ("noncausal", [7, 19, 3], None, None, 2, 32, False, 1),
    ("d128_long", [130, 77, 200], [5, 0, 9], [2, 1, 0], 1, 128, True, 1),
    # d = 256: the head dim of BASELINE configs 3-5 (C3 / C4 attention, H = 4 there; one head keeps the fixture small)
    ("d256_causal_plain", [40, 9, 75], None, None, 1, 256, True, 1),
    ("d256_ctx_targets", [66, 21, 90], [6, 2, 11], [3, 0, 2], 1, 256, True, 1),
]


def run(case, dtype):
    name, lengths, targets, ctx, H, d, causal, g = case
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2**31))
    off = torch.tensor([0] + list(np.cumsum(lengths)), dtype=torch.int64)
    L = int(off[-1])
    mk = lambda: torch.empty(L, H, d).uniform_(-1.0, 1.0, generator=gen).bfloat16().float()
    q, k, v = mk(), mk(), mk()
    dout = torch.empty(L, H, d).uniform_(0.0, 1.0, generator=gen).bfloat16().float()
    N = max(lengths)
    qq, kk, vv = [t.to(dtype).clone().requires_grad_(True) for t in (q, k, v)]
    out = pytorch_hstu_mha(
        max_seq_len=N, alpha=1.0 / d**0.5, q=qq, k=kk, v=vv, seq_offsets=off, causal=causal, dropout_pr=0.0,
        training=True, num_targets=None if targets is None else torch.tensor(targets, dtype=torch.int32),
        num_contextuals=None if ctx is None else torch.tensor(ctx, dtype=torch.int32),
        target_group_size=g, scaling_seqlen=N)
    out.backward(dout.to(dtype))
    return dict(q=q, k=k, v=v, dout=dout, off=off, N=N), out.detach().float(), qq.grad.float(), kk.grad.float(), vv.grad.float()


def main():
    blob = {}
    meta = []
    for case in CASES:
        name, lengths, targets, ctx, H, d, causal, g = case
        inp, o32, dq32, dk32, dv32 = run(case, torch.float32)
        _, o16, dq16, dk16, dv16 = run(case, torch.bfloat16)
        for kname, t in inp.items():
            if kname != "N":
                blob[f"{name}/{kname}"] = t.numpy()
        blob[f"{name}/targets"] = np.array(targets if targets is not None else [-1], np.int32)
        blob[f"{name}/ctx"] = np.array(ctx if ctx is not None else [-1], np.int32)
        blob[f"{name}/meta"] = np.array([H, d, int(causal), g, inp["N"]], np.int32)
        for kname, t in (("out", o32), ("dq", dq32), ("dk", dk32), ("dv", dv32), ("out_bf16", o16), ("dq_bf16", dq16),
                         ("dk_bf16", dk16), ("dv_bf16", dv16)):
            blob[f"{name}/{kname}"] = t.numpy().astype(np.float32)
        meta.append(name)
    blob["cases"] = np.array(meta)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB", meta)


if __name__ == "__main__":
    main()
