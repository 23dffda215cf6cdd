"""How does the all-D kernel's time follow the number of working wavefronts per pass?
    python tools/d_sweep.py [lib.so]
D = 66 / 72 / 78 / 84 / 90 are three passes of 22 / 24 / 26 / 28 / 30 disparities = 11 .. 15 working consumer wavefronts
(per SIMD 3/3/3/2 .. 4/4/4/3 beside the loader); D = 88 / 96 / 104 / 112 / 120 the same with four passes.  Kernel time per
pass and per disparity, with the volume (rank), without it (norank) and volume only (plain)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads
from connecting_the_dots_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from connecting_the_dots_amd import torchext as te
L = _lib.lib()
H, W, N = 432, 512, 16
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
x, _ = te.lcn(fr, 5, 0.05)
p, _ = te.lcn(pat, 5, 0.05)
p = p[0].contiguous()

def timed(fn, warm=300, reps=20):
    L.ctd_kernel_timing_enable(1)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    L.ctd_kernel_timing_collect(None, None)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    L.ctd_kernel_timing_enable(0)
    ms, cols = ctypes.c_double(0), ctypes.c_int(0)
    L.ctd_kernel_timing_collect(ctypes.byref(ms), ctypes.byref(cols))
    return ms.value

print("%4s %6s %6s %9s %9s %9s   %s" % ("D", "passes", "waves", "rank ms", "norank", "plain", "per pass (rank / norank / plain), us"))
for D in (66, 72, 78, 84, 90, 88, 96, 104, 112, 120, 110, 128, 130, 150):
    n_pass = -(-D // 30)
    dgs = 2 * -(-(-(-D // 2)) // n_pass)
    a = timed(lambda: te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast"))
    b = timed(lambda: te.xcorrvol_argmax(x, p, D, 9, algo="fast"))
    c = timed(lambda: te.xcorrvol_batch(x, p, D, 9, algo="fast"))
    print("%4d %6d %6d %9.4f %9.4f %9.4f   %.1f / %.1f / %.1f" % (D, n_pass, dgs // 2, a, b, c, a / n_pass * 1e3, b / n_pass * 1e3,
                                                             c / n_pass * 1e3), flush=True)
