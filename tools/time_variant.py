"""Times the ranked volume kernel of an experimental build of the library (tools/variants/*.so, never shipped):
    python tools/time_variant.py [path/to/lib.so ...]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads

def run(path):
    from connecting_the_dots_amd import _lib
    _lib._lib = None
    if not hasattr(_lib, "_IN_TREE"):
        _lib._IN_TREE = _lib.LIB_PATH
    _lib.LIB_PATH = path or _lib._IN_TREE
    from connecting_the_dots_amd import torchext as te
    L = _lib.lib()
    H, W, D, N = 432, 512, 128, 16
    fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
    pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
    x, _ = te.lcn(fr, 5, 0.05)
    p, _ = te.lcn(pat, 5, 0.05)
    p = p[0].contiguous()
    L.ctd_kernel_timing_enable(1)
    for _ in range(int(os.environ.get("CTD_WARM_CALLS", "100"))):                # ~50 ms of the same work before timing: the card's clocks need it
        te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast")
    torch.cuda.synchronize()
    L.ctd_kernel_timing_collect(None, None)
    t0 = time.perf_counter()
    for _ in range(20):
        te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    L.ctd_kernel_timing_enable(0)
    ms, cols = ctypes.c_double(0), ctypes.c_int(0)
    n = L.ctd_kernel_timing_collect(ctypes.byref(ms), ctypes.byref(cols))
    L.ctd_kernel_timing_enable(1)
    for _ in range(60):
        te.xcorrvol_batch(x, p, D, 9, algo="fast")
    torch.cuda.synchronize()
    L.ctd_kernel_timing_collect(None, None)
    for _ in range(20):
        te.xcorrvol_batch(x, p, D, 9, algo="fast")
    torch.cuda.synchronize()
    L.ctd_kernel_timing_enable(0)
    ms2 = ctypes.c_double(0)
    L.ctd_kernel_timing_collect(ctypes.byref(ms2), ctypes.byref(cols))
    print("%-28s ranked volume kernel %.4f ms (%d launches)  argmax call %.4f ms   plain volume kernel %.4f ms" % (
        os.path.basename(path or "in-tree"), ms.value, n, dt * 1e3, ms2.value), flush=True)

for _ in range(3):                      # interleaved repetitions: box-to-box and run-to-run spread shows up
    for p in (sys.argv[1:] or [""]):
        run(p)
