"""LCN kernel in a loop (profiling target: rocprofv3 --kernel-trace --stats -- python tools/prof_lcn.py [variant.so])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from connecting_the_dots_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from connecting_the_dots_amd import torchext as te
from tests import workloads
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, 432, 512) for i in range(16)])).cuda()
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for _ in range(300):
    y, s = te.lcn(fr, 5, 0.05)
    if _ % 4 == 0:
        big.fill_(1)                 # 512 MB through the caches now and then: the bench step's LCN starts cold
torch.cuda.synchronize()
