"""Build recipe of libctd_hip.so (hipcc, gfx950 only, in-tree so that gpurun ships it).

    python -m connecting_the_dots_amd.build [--force]

-ffp-contract=off is global: the reference-order kernels must not fuse multiply-adds
(the parity anchor is the reference's FMA-free CPU build); kernels that want an FMA
call fmaf() explicitly.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libctd_hip.so")
HEADER = os.path.join(os.path.dirname(PKG), "include", "ctd_hip.h")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
