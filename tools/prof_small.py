import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""Profiling target: only the 16-frame LCN (and a fast volume call for its pre-pass)."""
import torch
from connecting_the_dots_amd import torchext as te
torch.manual_seed(0)
a = torch.rand(16, 1, 432, 512, device="cuda")
for _ in range(3):
    x, _ = te.lcn(a, 5, 0.05)
torch.cuda.synchronize()
