"""Kernel-time table for the training-side ops at BASELINE sizes (rocprofv3 --kernel-trace --stats wraps this)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from connecting_the_dots_amd import torchext as te
B, H, W = 16, 432, 512
torch.manual_seed(0)
es = torch.rand(B, 1, H, W, device="cuda"); ta = torch.rand(B, 1, H, W, device="cuda")
def timeit(name, fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-40s %8.3f ms  %8.1f Mpix/s" % (name, dt * 1e3, B * H * W / dt / 1e6), flush=True)
for algo in ("exact", "fast"):
    for ty in ("mse", "sad", "census_mse", "census_sad"):
        e = es.clone().requires_grad_(True)
        timeit("photometric fwd %s %s" % (ty, algo), lambda: te.photometric_loss(e.detach(), ta, 9, ty, 0.5, algo=algo))
        out = te.photometric_loss(e, ta, 9, ty, 0.5, algo=algo); go = torch.rand_like(out)
        timeit("photometric bwd %s %s" % (ty, algo), lambda: torch.autograd.grad(out, e, go, retain_graph=True))
timeit("lcn", lambda: te.lcn(es, 5, 0.05))
d = (torch.rand(B, 1, H, W, device="cuda") * 60).requires_grad_(True); edge = torch.rand(B, 1, H, W, device="cuda")
timeit("disparity_loss fwd", lambda: te.disparity_loss(d.detach(), edge))
l = te.disparity_loss(d, edge)
timeit("disparity_loss bwd", lambda: torch.autograd.grad(l, d, retain_graph=True))
timeit("disp_to_depth fwd", lambda: te.disp_to_depth(d.detach(), 567.6 * 0.075))
timeit("costvol census_sad D=128 (1 frame)", lambda: te.costvol(es[0, 0], ta[0, 0], 128, 9, "census_sad", 0.5), n=3)
pattern = torch.randn(1, 3, H, W, device="cuda"); std = 0.05 + torch.rand(B, 1, H, W, device="cuda")
for algo in ("exact", "fast"):
    mod = te.RectifiedPatternSimilarityLoss(H, W, pattern, algo=algo)
    dd = (torch.rand(B, 1, H, W, device="cuda") * 60).requires_grad_(True)
    def step():
        v, _ = mod(dd, ta, std); v.backward(); dd.grad = None
    timeit("pattern loss fwd+bwd %s" % algo, step)
# N2: the four pyramid levels of a training step (B = 8), per-level calls vs one launch each way
Bm = 8; lv = []
for s in range(4):
    Hs, Ws = 480 >> s, 640 >> s
    lv.append((torch.randn(1, 1, Hs, Ws, device="cuda"), torch.randn(Bm, 1, Hs, Ws, device="cuda"),
               0.05 + torch.rand(Bm, 1, Hs, Ws, device="cuda"), (torch.rand(Bm, 1, Hs, Ws, device="cuda") * (60 >> s)).requires_grad_(True)))
def per_level():
    tot = 0
    for p, im, st, d in lv:
        v, _, _ = te.pattern_loss(d, im, st, p, "census_sad", 0.5); tot = tot + v
    tot.backward()
    for _, _, _, d in lv: d.grad = None
def one_launch():
    vals, _, _ = te.pattern_loss_multi([l[3] for l in lv], [l[1] for l in lv], [l[2] for l in lv], [l[0] for l in lv])
    vals.sum().backward()
    for _, _, _, d in lv: d.grad = None
B = Bm; H, W = 480, 640
timeit("4-level pattern loss, per level", per_level)
timeit("4-level pattern loss, one launch", one_launch)
# A10: two-view geometric loss (both directions), 15 consecutive pairs of 432x512 depth maps
import numpy as np
B = 15; H, W = 432, 512
K = torch.tensor([[567.6, 0, 324.7], [0, 570.2, 250.1], [0, 0, 1]], device="cuda"); Ki = torch.linalg.inv(K.double()).float()
depth = (0.5 + torch.rand(16, 1, H, W, device="cuda") * 3).requires_grad_(True)
R = torch.eye(3, device="cuda").repeat(16, 1, 1).contiguous(); t = (torch.randn(16, 3, device="cuda") * 0.02).contiguous()
gl = te.ProjectionDepthSimilarityLoss(K, Ki, H, W, clamp=0.1)
def geo_fwd():
    with torch.no_grad(): gl(depth[:-1].contiguous(), depth[1:].contiguous(), R[:-1].contiguous(), t[:-1].contiguous(), R[1:].contiguous(), t[1:].contiguous())
def geo_fb():
    v = gl(depth[:-1].contiguous(), depth[1:].contiguous(), R[:-1].contiguous(), t[:-1].contiguous(), R[1:].contiguous(), t[1:].contiguous()); v.backward(); depth.grad = None
timeit("geometric loss fwd (2 directions)", geo_fwd)
timeit("geometric loss fwd+bwd", geo_fb)
