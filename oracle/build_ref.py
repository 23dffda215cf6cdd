"""Recipe: compile the reference's own CPU binding into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path imports this.

Compiles /root/reference/torchext/ext/ext_cpu.cpp (which pulls in ext.h and
common.h from the same directory) *from where it lies* with g++ against the
installed torch headers.  No reference source is copied; the only output is
oracle/_ref/ext_cpu*.so (git-ignored, but shipped to the GPU box by gpurun so
bench.py can time the real reference CPU path there).

The reference builds the same file through torch.utils.cpp_extension with no
extra flags (torchext/setup.py:16-18); we add -O3 as the survey did and keep
x86-64 baseline codegen (no -march=native, no -mfma), i.e. no FMA contraction,
which is what makes this build the bit-exact parity anchor.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/torchext/ext/ext_cpu.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
MOD_NAME = "ctd_ref_ext_cpu"


def out_path():
    return os.path.join(OUT_DIR, MOD_NAME + ".so")


def build(force=False, verbose=True):
    """Build oracle/_ref/ctd_ref_ext_cpu.so.  Returns the path, or None when the
    reference tree is absent (GPU box) and no prebuilt file exists."""
    so = out_path()
    if not os.path.exists(REF_SRC):
        return so if os.path.exists(so) else None
    if os.path.exists(so) and not force and os.path.getmtime(so) >= os.path.getmtime(REF_SRC):
        return so
    from torch.utils import cpp_extension as ce
    import torch
    os.makedirs(OUT_DIR, exist_ok=True)
    incs = ce.include_paths() + [sysconfig.get_paths()["include"]]
    libdir = ce.library_paths()[0]
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-w",
           "-DTORCH_EXTENSION_NAME=" + MOD_NAME,
           "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for i in incs:
        cmd += ["-isystem", i]
    cmd += [REF_SRC, "-o", so, "-L" + libdir, "-Wl,-rpath," + libdir,
            "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return so


RENDER_DIR = "/root/reference/renderer/render"
RENDER_OUT = os.path.join(OUT_DIR, "ctd_ref_render.so")
RENDER_DRIVER = os.path.join(HERE, "ref_render_driver.cpp")


def build_render(verbose=True):
    """The reference's CPU renderer (renderer/render/*.h, render_cpu.cpp) behind oracle/ref_render_driver.cpp,
    compiled with the reference's own flags (renderer/setup.py: -O3 -std=c++11; no FMA on baseline x86-64).
    Returns the path, or None where the reference is absent."""
    if not os.path.exists(os.path.join(RENDER_DIR, "render_cpu.cpp")):
        return RENDER_OUT if os.path.exists(RENDER_OUT) else None
    if os.path.exists(RENDER_OUT) and os.path.getmtime(RENDER_OUT) >= os.path.getmtime(RENDER_DRIVER):
        return RENDER_OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O3", "-std=c++11", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-w", "-I", RENDER_DIR,
           RENDER_DRIVER, "-o", RENDER_OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return RENDER_OUT


def load_render():
    import ctypes
    path = build_render(verbose=False)
    if path is None:
        raise RuntimeError("reference renderer not available (no /root/reference and no prebuilt oracle/_ref)")
    return ctypes.CDLL(path)


def load():
    """Import the compiled reference module (needs torch imported first)."""
    import importlib.util
    import torch  # noqa: F401  (registers libtorch symbols)
    so = build(verbose=False)
    if so is None or not os.path.exists(so):
        return None
    spec = importlib.util.spec_from_file_location(MOD_NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print("reference CPU binding:", p)
