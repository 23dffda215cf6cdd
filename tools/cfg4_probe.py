import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from connecting_the_dots_amd import torchext as te
from tests import workloads
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frame = workloads.uniform_frame(77, H, W); pat = workloads.syn_dot_pattern(H, W, seed=42)[None]
x, _ = te.lcn(torch.from_numpy(frame[None]).cuda(), 5, 0.05); p, _ = te.lcn(torch.from_numpy(pat[None]).cuda(), 5, 0.05)
x, p = x[0].contiguous(), p[0].contiguous()
exact = te.xcorrvol_batch(x[None], p, D, 9, algo="exact")[0]
fast = te.xcorrvol_batch(x[None], p, D, 9, algo="fast")[0]
err = (fast - exact).abs(); bad = err > exact.abs() * 1e-5 + 1e-6
print("bad count", bad.sum().item(), "of", bad.numel(), "max err", err.max().item(), "nan", torch.isnan(fast).sum().item())
nz = bad.nonzero()
if len(nz):
    print("d range", nz[:, 0].min().item(), nz[:, 0].max().item(), "h range", nz[:, 1].min().item(), nz[:, 1].max().item(), "w range", nz[:, 2].min().item(), nz[:, 2].max().item())
    print("bad by d (first 40 nonzero):", [(d, int(c)) for d, c in enumerate(bad.sum((1, 2)).tolist()) if c][:40])
    wb = bad.sum((0, 1)); print("bad by w blocks of 64:", wb.view(-1, 64).sum(1).tolist())
    hb = bad.sum((0, 2)); print("bad by h blocks of 64:", hb.view(-1, 64).sum(1).tolist())
    print("examples", nz[:5].tolist(), [ (fast[tuple(i)].item(), exact[tuple(i)].item()) for i in nz[:5]])
