"""Config-2 step, three ways: lcn(fast) + xcorrvol_argmax | lcn(exact) + xcorrvol_argmax | fused lcn_xcorrvol_argmax
(exact / fast LCN sums).  Wall time per step over 200 steps after 300 untimed ones; run under rocprofv3 --kernel-trace for
the per-kernel split (tools/kstats.py)."""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import _lib  # noqa: E402
if os.environ.get('CTD_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['CTD_LIB'])
from connecting_the_dots_amd import torchext as te  # noqa: E402
from tests import workloads  # noqa: E402

H, W, D, N = 432, 512, 128, 16
modes = sys.argv[1:] or ["unfused_fast", "unfused_exact", "fused_exact", "fused_fast"]
x = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
pat = te.lcn(torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda(), 5, 0.05)[0][0].contiguous()
prep = te.prepare_pattern(pat, N, D, 9)


def step(mode):
    if mode == "unfused_fast":
        y, _ = te.lcn(x, 5, 0.05, algo="fast")
        return te.xcorrvol_argmax(y, pat, D, 9, return_volume=True, prepared=prep)
    if mode == "unfused_exact":
        y, _ = te.lcn(x, 5, 0.05, algo="exact")
        return te.xcorrvol_argmax(y, pat, D, 9, return_volume=True, prepared=prep)
    return te.lcn_xcorrvol_argmax(x, pat, D, 9, 5, 0.05, return_volume=True, lcn_algo=mode.split("_")[1], prepared=prep)


for mode in modes:
    for _ in range(300):
        held = step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        held = step(mode)
    torch.cuda.synchronize()
    print("%-14s %.4f ms per step" % (mode, (time.perf_counter() - t0) / 200 * 1e3), flush=True)
