"""Does replaying the config-2 step as ONE captured graph beat launching its five kernels one by one?  (probe)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te
from tests import workloads
H, W, D, N = 432, 512, 128, 16
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda().reshape(N, 1, H, W)
pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
p = te.lcn(pat, 5, 0.05)[0][0].contiguous()
prep = te.prepare_pattern(p, N, D, 9)

def step():
    x, _ = te.lcn(fr, 5, 0.05)
    return te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast", prepared=prep)

def timeit(fn, n):
    for _ in range(600):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

ref = step()
print("eager  %.4f ms per step" % timeit(step, 200), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
g.replay(); torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(out, ref)), "graph replay differs"
print("graph  %.4f ms per step" % timeit(g.replay, 200), flush=True)
print("eager  %.4f ms per step" % timeit(step, 200), flush=True)
