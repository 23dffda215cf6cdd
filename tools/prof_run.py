import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""Runs one hot-path kernel family a few times (profiling target for rocprofv3 --pmc passes):
    python tools/prof_run.py [fast|exact|rank|norank] [frames] [variant.so]
    (rank = the all-D kernel, volume + ranking; norank = the same without a volume; variant.so: a tools/variants build)"""
import sys, torch
from connecting_the_dots_amd import _lib
if len(sys.argv) > 3:
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
from connecting_the_dots_amd import torchext as te
algo = sys.argv[1] if len(sys.argv) > 1 else "fast"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, W, D = 432, 512, 128
torch.manual_seed(0)
a = torch.rand(N, 1, H, W, device="cuda"); b = torch.rand(1, 1, H, W, device="cuda")
a, _ = te.lcn(a, 5, 0.05); b, _ = te.lcn(b, 5, 0.05); b = b[0].contiguous()
for _ in range(3):
    if algo == "rank":
        v = te.xcorrvol_argmax(a, b, D, 9, return_volume=True, algo="fast")
    elif algo == "norank":
        v = te.xcorrvol_argmax(a, b, D, 9, algo="fast")
    else:
        v = te.xcorrvol_batch(a, b, D, 9, algo=algo)
torch.cuda.synchronize()
