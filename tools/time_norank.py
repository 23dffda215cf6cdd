"""All-D kernel time by library variant, three modes (volume + ranking, ranking only, volume only), sustained clocks:
    python tools/time_norank.py [lib.so ...]     ("" = the in-tree library)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads

def run(path, D=128):
    from connecting_the_dots_amd import _lib
    _lib._lib = None
    if not hasattr(_lib, "_IN_TREE"):
        _lib._IN_TREE = _lib.LIB_PATH
    _lib.LIB_PATH = os.path.abspath(path) if path else _lib._IN_TREE
    from connecting_the_dots_amd import torchext as te
    L = _lib.lib()
    H, W, N = 432, 512, 16
    fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
    pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
    x, _ = te.lcn(fr, 5, 0.05)
    p, _ = te.lcn(pat, 5, 0.05)
    p = p[0].contiguous()
    out = []
    for fn in (lambda: te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast"),
               lambda: te.xcorrvol_argmax(x, p, D, 9, algo="fast"),
               lambda: te.xcorrvol_batch(x, p, D, 9, algo="fast")):
        L.ctd_kernel_timing_enable(1)
        for _ in range(int(os.environ.get("CTD_WARM_CALLS", "600"))):
            fn()
        torch.cuda.synchronize()
        L.ctd_kernel_timing_collect(None, None)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        L.ctd_kernel_timing_enable(0)
        ms, cols = ctypes.c_double(0), ctypes.c_int(0)
        L.ctd_kernel_timing_collect(ctypes.byref(ms), ctypes.byref(cols))
        out.append(ms.value)
    print("%-26s D %3d   volume+rank %.4f ms   rank only %.4f   volume only %.4f" % (os.path.basename(path or "in-tree"), D, *out), flush=True)

Ds = [int(v) for v in os.environ.get("CTD_DS", "128").split(",")]
for _ in range(2):
    for D in Ds:
        for p in (sys.argv[1:] or [""]):
            run(p, D)
