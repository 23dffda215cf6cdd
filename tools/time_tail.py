"""Profiling target: the ranked argmax call of the bench workload, 60 calls (run under rocprofv3 --kernel-trace --stats):
    python tools/time_tail.py [variant.so]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from connecting_the_dots_amd import torchext as te
from tests import workloads
H, W, D, N = 432, 512, 128, 16
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
x, _ = te.lcn(fr, 5, 0.05)
p, _ = te.lcn(pat, 5, 0.05)
p = p[0].contiguous()
prepared = te.prepare_pattern(p, N, D, 9)                # as the bench step: the pattern half of the pre-pass once
for _ in range(200):
    te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast", prepared=prepared)
torch.cuda.synchronize()
