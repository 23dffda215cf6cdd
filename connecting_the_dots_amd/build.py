"""Build recipe of libctd_hip.so (hipcc, gfx950 only, in-tree so that gpurun ships it).

    python -m connecting_the_dots_amd.build [--force]

Every csrc/*.hip is compiled to its own object under build/obj (in parallel, only when the source or a header is
newer than the object) and the objects are linked into the one shared library.  No relocatable device code is
needed: device functions shared between files live in headers.

-ffp-contract=off is global: the reference-order kernels must not fuse multiply-adds
(the parity anchor is the reference's FMA-free CPU build); kernels that want an FMA
call fmaf() explicitly.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libctd_hip.so")
OBJ_DIR = os.path.join(os.path.dirname(PKG), "build", "obj")
HEADER = os.path.join(os.path.dirname(PKG), "include", "ctd_hip.h")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HEADER), "*.h"))


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, sources() + _headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = _headers()
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + hdrs)]

    def compile_one(src):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in sources()] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
