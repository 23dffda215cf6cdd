"""How much do the kernel-timing events cost a bench step?  Alternates regions of K steps with the hooks on / off,
and with the previous step's outputs held while the next step runs (two volumes alternate) or dropped."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from connecting_the_dots_amd import _lib, torchext as te   # noqa: E402
from tests import workloads                                 # noqa: E402

H, W, D, N = 432, 512, 128, 16
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
p = te.lcn(pat, 5, 0.05)[0][0].contiguous()
L = _lib.lib()


def step():
    x, _ = te.lcn(fr, 5, 0.05)
    return te.xcorrvol_argmax(x, p, D, 9, return_volume=True)


for _ in range(20):
    step()
torch.cuda.synchronize()
for rep in range(3):
  for keep in (1, 0):
    for hooks in (1, 0):
        L.ctd_kernel_timing_enable(hooks)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            if keep:
                held = step()
            else:
                step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20 * 1e3
        L.ctd_kernel_timing_enable(0)
        a, c = ctypes.c_double(0), ctypes.c_int(0)
        n = L.ctd_kernel_timing_collect(ctypes.byref(a), ctypes.byref(c))
        print("keep=%d hooks=%d  %.4f ms/step  kernel avg %.4f ms over %d" % (keep, hooks, dt, a.value, n), flush=True)
