"""Per-phase cycle breakdown of the d = 256 attention kernels -- hstu_fwd_kernel (the one-kind LDS-DMA forward of round 3 was removed in round 5), the dK pass of
the backward (--bwd).  Needs a library whose hstu_attn.hip was compiled with -DHSTU_TIMING=1:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHSTU_TIMING=1 -c recsys-examples_amd/csrc/hstu_attn.hip -o /tmp/h.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o recsys-examples_amd/lib/librecsys_amd_tim.so \
        $(ls recsys-examples_amd/lib/obj/*.o | grep -v hstu_attn.o) /tmp/h.o
    MI355_LIB=$PWD/recsys-examples_amd/lib/librecsys_amd_tim.so python tools/hstu_phase_cycles.py [--dma | --bwd] [--seqlen L --batch B]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recsys-examples_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hstu import hstu_varlen_bwd, hstu_varlen_fwd
import mi355_native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--seqlen", type=int, default=512)
ap.add_argument("--heads", type=int, default=4); ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--pc", action="store_true", help="the two-waves-per-SIMD forward (hstu_fwd_pc_kernel: S waves / O waves; the default forward)")
ap.add_argument("--q2", action="store_true", help="the 64-rows-per-wave forward (hstu_fwd_q2_kernel, the default forward since round 4)")
ap.add_argument("--bwdpc", action="store_true", help="the S-wave / K-wave dK pass of the backward (hstu_bwd_kv_pc_kernel)")
ap.add_argument("--bwd", action="store_true", help="the dK pass of the backward (hstu_bwd_kv_kernel, exchange mode) instead of the forward")
a = ap.parse_args()
if a.q2:
    a.pc = True
    os.environ["MI355_HSTU_FWD"] = "1"
elif a.pc:
    os.environ["MI355_HSTU_FWD"] = "3"   # 32-row waves, unpaired: the stamps live in hstu_fwd_pc_kernel
elif not a.bwd:
    os.environ.setdefault("MI355_HSTU_FWD", "5")
dev = torch.device("cuda")
T = a.batch * a.seqlen
cu = torch.arange(0, T + 1, a.seqlen, dtype=torch.int32, device=dev)
q, k, v = (torch.empty(T, a.heads, a.dim, device=dev).uniform_(-1, 1).bfloat16() for _ in range(3))
for _ in range(3):
    if a.bwd or a.bwdpc:
        hstu_varlen_bwd(q, q, k, v, cu, a.seqlen, a.seqlen, None, None, 1, True, 1.0 / a.dim ** 0.5)
    else:
        hstu_varlen_fwd(q, k, v, cu, a.seqlen, a.seqlen, None, None, 1, True, 1.0 / a.dim ** 0.5)
torch.cuda.synchronize()
nblk = a.heads * a.batch * ((a.seqlen + 127) // 128)
n = min(nblk * 4, 65536)
buf = np.zeros((65536, 8), np.uint64)
lib = ctypes.CDLL(mi355_native.LIB_PATH)
lib.mi355_hstu_dbg_dump.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.mi355_hstu_dbg_dump(buf.ctypes.data, buf.nbytes) == 0
if a.bwdpc:
    n = min(nblk * 8, 65536)
    dd = buf[:n].astype(np.float64)
    role_of = (buf[:n, 6] >> np.uint64(32)).astype(np.int64)
    dd[:, 6] = (buf[:n, 6] & np.uint64(0xffffffff)).astype(np.float64)
    for role, nm0, names in ((0, 'S waves (S, dP, elementwise)', ['barrier', None, None, None, 'gemm S + dP', 'elementwise + stores']), (1, 'K waves (DMA, dK)', ['wait own DMA', 'barrier', 'DMA issue', None, None, 'gemm dK'])):
        d = dd[role_of == role]
        tiles, tot = d[:, 6], d[:, 7]
        print(f'{nm0}: waves {len(d)}  steps/wave avg {tiles.mean():.2f}  wave lifetime avg {tot.mean():.0f} cyc  max {tot.max():.0f}')
        for i, nm in enumerate(names):
            if nm is None: continue
            print(f'  {nm:26s} {d[:, i].sum() / tiles.sum():8.0f} cyc per computed step   ({100 * d[:, i].sum() / tot.sum():5.1f} % of wave lifetime)')
        print(f"  {'other':26s} {(tot.sum() - d[:, :6].sum()) / tiles.sum():8.0f} cyc per computed step   ({100 * (tot.sum() - d[:, :6].sum()) / tot.sum():5.1f} %)")
    sys.exit(0)
if a.pc:
    n = min(nblk * 8, 65536)
    dd = buf[:n].astype(np.float64)
    role_of = (buf[:n, 6] >> np.uint64(32)).astype(np.int64)
    dd[:, 6] = (buf[:n, 6] & np.uint64(0xffffffff)).astype(np.float64)
    for role, nm0 in ((0, "S waves (GEMM 1 + SiLU -> P)"), (1, "O waves (DMA + GEMM 2)")):
        d = dd[role_of == role]
        tiles, tot = d[:, 6], d[:, 7]
        print(f"{nm0}: waves {len(d)}  tiles/wave avg {tiles.mean():.2f}  wave lifetime avg {tot.mean():.0f} cyc  max {tot.max():.0f}")
        names = (["barrier .. gemm1 start", "barrier", None, "tile end .. next barrier", "gemm1", "silu + P hand-off"] if role == 0 and a.q2 else
                 ["wait own DMA", "barrier", "DMA issue", "gemm1 sub-tile 1 + silu 0", "gemm1 sub-tile 0", "silu sub-tile 1"] if role == 0
                 else ["wait own DMA", "barrier", "DMA issue", None, None, "P read + gemm2"])
        for i, nm in enumerate(names):
            if nm is None:
                continue
            print(f"  {nm:26s} {d[:, i].sum() / tiles.sum():8.0f} cyc per computed tile   ({100 * d[:, i].sum() / tot.sum():5.1f} % of wave lifetime)")
        used = d[:, :6].sum()
        print(f"  {'other':26s} {(tot.sum() - used) / tiles.sum():8.0f} cyc per computed tile   ({100 * (tot.sum() - used) / tot.sum():5.1f} %)")
    sys.exit(0)
d = buf[:n].astype(np.float64)
tiles = d[:, 6]
names = (["barrier1", "fetch (load + wait)", "commit x2 images", "barrier2", "gemm S + dP", "silu' + P/dS stores"] if a.bwd
         else ["barrier1", "commit", "barrier2", "fetch-issue", "gemm1", "silu+gemm2"])
tot = d[:, 7]
print(f"waves {n}  tiles/wave avg {tiles.mean():.2f}  wave lifetime avg {tot.mean():.0f} cyc  max {tot.max():.0f}")
loop_iters = np.maximum(tiles, 1)
for i, nm in enumerate(names):
    print(f"  {nm:22s} {d[:, i].sum() / tiles.sum():8.0f} cyc per computed tile   ({100 * d[:, i].sum() / tot.sum():5.1f} % of wave lifetime)")
print(f"  {'gemm dK + rest' if a.bwd else 'other':12s} {(tot.sum() - d[:, :6].sum()) / tiles.sum():8.0f} cyc per computed tile   ({100 * (tot.sum() - d[:, :6].sum()) / tot.sum():5.1f} %)")
