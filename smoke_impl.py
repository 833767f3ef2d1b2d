"""One tiny end-to-end pass of the DynamicEmb hot path on cuda:0, checked against the oracle
(used by __graft_entry__.smoke())."""
import numpy as np
import torch


def run():
    import dynamicemb_extensions as ext
    from dynamicemb.scored_hashtable import LinearBucketTable, ScoreArg, ScoreSpec
    from oracle import oracle as orc

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    # the reference's 11-key / 4-feature / batch-2 fixture (test_batched_dynamic_embedding_tables_v2.py:1517-1522)
    indices = np.array([0, 1, 12, 64, 8, 12, 15, 2, 7, 105, 0], np.int64)
    offsets = np.array([0, 2, 3, 5, 6, 8, 10, 10, 11], np.int64)
    fo = np.array([0, 2, 3, 4], np.int64)
    B, D, T = 2, 8, 3
    t = lambda a: torch.from_numpy(a).to(dev)
    rng_t = ext.get_table_range(t(offsets), t(fo))
    num, uk, rev, to, _ = ext.segmented_unique_cuda(t(indices), rng_t, T, None)
    nu = int(num.item())
    tids = ext.expand_table_ids_cuda(to, nu)
    table = LinearBucketTable([2048] * T, [ScoreSpec("s", ext.ScorePolicy.ASSIGN)], device=dev)
    slots = table.insert(uk[:nu].contiguous(), tids, ScoreArg("s", torch.ones(nu, dtype=torch.int64, device=dev).view(torch.uint64)))
    vals = [torch.zeros(2048, D, device=dev) for _ in range(T)]
    tptr = torch.tensor([v.data_ptr() for v in vals], dtype=torch.int64, device=dev)
    vd = torch.full((T,), D, dtype=torch.int64, device=dev)
    addr = ext.row_addresses(slots, tids, tptr, vd, 4)
    ext.init_rows(4, (0, 0, 0, 0), 0, 0.0, uk[:nu].contiguous(), addr, torch.float32, D, D)  # DEBUG: row = key % 100000
    out = torch.empty(B, 4 * D, device=dev)
    ext.gather_embedding_pooled(None, out, rev, t(offsets), 0, 4 * D, B, max_D=D, row_addr=addr, src_dtype=torch.float32)
    # oracle
    ouk, orev, oto, _ = orc.segmented_unique(indices, orc.get_table_range(offsets, fo, B))
    exp = orc.gather_pooled(orc.debug_init(ouk, D), orev, offsets, B, 0)
    torch.cuda.synchronize()
    assert nu == ouk.size and (rev.cpu().numpy() == orev).all(), "unique mismatch"
    assert (out.cpu().numpy() == exp).all(), "pooled output mismatch"
    # backward: SGD in place, closed form w -= lr * sum(grad rows)
    g = torch.ones(B, 4 * D, device=dev)
    ptr_t, csr, hot = ext.group_by_unique(rev, nu, t(offsets), dim=D)
    ext.backward_fused(ptr_t, csr, indices.size, nu, g, B, D, 0, t(offsets), None, addr, torch.float32, 1, lr=0.5, hot=hot)
    out2 = torch.empty(B, 4 * D, device=dev)
    ext.gather_embedding_pooled(None, out2, rev, t(offsets), 0, 4 * D, B, max_D=D, row_addr=addr, src_dtype=torch.float32)
    cnt = np.bincount(orev, minlength=nu).astype(np.float32)
    exp2 = orc.gather_pooled(orc.debug_init(ouk, D) - 0.5 * cnt[:, None], orev, offsets, B, 0)
    torch.cuda.synchronize()
    assert np.allclose(out2.cpu().numpy(), exp2), "backward/SGD mismatch"
    print("smoke ok: unique/insert/init/pool/backward match the oracle on the 11-key fixture")
    _hstu_smoke(dev)


def _hstu_smoke(dev):
    """path B: one tiny jagged HSTU attention forward + backward (targets, causal) against the float64 oracle"""
    from hstu import hstu_attn_varlen_func
    from oracle import hstu_oracle as ho

    _hstu_case(dev, 64, np.array([37, 5, 130]), ho, hstu_attn_varlen_func)
    # head dim 256 = the BASELINE configs' attention: the two-waves-per-SIMD kernels of round 4 (S-wave / O-wave forward,
    # S-wave / K-wave dK pass, DMA-staged dV / dQ passes)
    _hstu_case(dev, 256, np.array([200, 5, 333]), ho, hstu_attn_varlen_func)
    print("smoke ok: hstu_attn_varlen_func forward/backward match the oracle (d = 64 and d = 256)")


def _hstu_case(dev, d, lengths, ho, hstu_attn_varlen_func):
    H = 2
    N = int(lengths.max())
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    q, k, v = (torch.empty(T, H, d, device=dev).uniform_(-1, 1).bfloat16().requires_grad_() for _ in range(3))
    dout = torch.empty(T, H, d, device=dev).uniform_(0, 1).bfloat16()
    tg = np.array([3, 1, 7])
    cu = torch.from_numpy(off.astype(np.int32)).to(dev)
    out = hstu_attn_varlen_func(q, k, v, cu, cu, None, None, N, N, N, None,
                                torch.from_numpy(tg.astype(np.int32)).to(dev), target_group_size=1, window_size=(-1, 0),
                                alpha=1.0 / d ** 0.5)
    out.backward(dout)
    qn, kn, vn, dn = (x.detach().float().cpu().numpy() for x in (q, k, v, dout))
    ref = ho.hstu_attn_fwd(qn, kn, vn, off, 1.0 / d ** 0.5, N, True, tg, None, 1)
    dq, dk, dv = ho.hstu_attn_bwd(dn, qn, kn, vn, off, 1.0 / d ** 0.5, N, True, tg, None, 1)
    for got, want, tol in ((out, ref, 6e-3), (q.grad, dq, 1.2e-2), (k.grad, dk, 1.2e-2), (v.grad, dv, 1.2e-2)):
        err = np.abs(got.detach().float().cpu().numpy() - want).max()
        assert err <= tol * np.abs(want).max() + 1e-6, f"hstu smoke mismatch {err} (d = {d})"
