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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math" if False else "-O3"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

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
    print(build(force="--force" in sys.argv))
