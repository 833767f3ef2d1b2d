"""Builds librecsys_amd.so (gfx950) in-tree with hipcc.  No cmake, no torch extension machinery:
the library is a plain C-ABI shared object (include/recsys_amd.h) loaded with ctypes.

    python recsys-examples_amd/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librecsys_amd.so")
OBJDIR = os.path.join(LIBDIR, "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# precise math on purpose: the reference builds its optimizer kernels with --use_fast_math (__powf); here powf / expf / sqrtf
# keep their IEEE forms so that the fused optimizers match the oracle (and torch.optim) to the last bit in the parity tests
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=(), only=()) -> str:
    """variant / defines: a side build for A/B sweeps and profiling (lib/librecsys_amd_<variant>.so, objects under
    lib/obj_<variant>/), selected at run time with MI355_LIB; the default build is the product."""
    global OBJDIR, LIB
    objdir_saved, lib_saved = OBJDIR, LIB
    if variant:
        OBJDIR = os.path.join(LIBDIR, "obj_" + variant)
        LIB = os.path.join(LIBDIR, f"librecsys_amd_{variant}.so")
    try:
        return _build(force, verbose, list(defines), tuple(only), objdir_saved)
    finally:
        OBJDIR, LIB = objdir_saved, lib_saved


def _build(force, verbose, defines, only=(), base_objdir=None) -> str:
    """only: sources a variant build recompiles with `defines`; every other object is taken from the default build"""
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if only and src not in only:
            objs.append(os.path.join(base_objdir, src.replace(".hip", ".o")))
            continue
        objs.append(o)
        # (a source may be a second translation unit of another one: hstu_attn_f16.hip includes hstu_attn.hip)
        inc = [os.path.join(CSRC, m) for m in __import__("re").findall(r'#include "([^"]+\.hip)"', open(s).read())]
        if force or _stale(o, [s] + inc + headers):
            jobs.append([HIPCC] + FLAGS + defines + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd[-3:]), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    _only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else ()
    print(build(force="--force" in sys.argv, variant=_variant, defines=[a for a in sys.argv[1:] if a.startswith("-D")], only=_only))
