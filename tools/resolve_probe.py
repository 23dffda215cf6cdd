import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from connecting_the_dots_amd import torchext as te
import bench
eps = float(sys.argv[1])
frames, pattern = bench.make_inputs(16, 0, torch.device("cuda"))
pl, _ = te.lcn(pattern, 5, 0.05); pl = pl[0].contiguous()
x, _ = te.lcn(frames, 5, 0.05)
for _ in range(4):
    idx, best = te.xcorrvol_argmax(x, pl, 128, 9, algo="fast", rerank_eps=eps)
torch.cuda.synchronize()
