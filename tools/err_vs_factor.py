"""Empirical error model of the fast NCC: err / (eps32 * sqrt(Fa*Fb)) where F = 1 + n*m'^2/var per window."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from connecting_the_dots_amd import _lib
if os.environ.get('CTD_VARIANT'):
    _lib.LIB_PATH = os.path.abspath(os.environ['CTD_VARIANT'])
from connecting_the_dots_amd import torchext as te
from tests import workloads
H, W, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "lcn"
frame = workloads.uniform_frame(77, H, W); pat = workloads.syn_dot_pattern(H, W, seed=42)[None]
if kind == "lcn":
    x, _ = te.lcn(torch.from_numpy(frame[None]).cuda(), 5, 0.05); p, _ = te.lcn(torch.from_numpy(pat[None]).cuda(), 5, 0.05)
    x, p = x[0].contiguous(), p[0].contiguous()
else:   # raw intensities with offset, smooth gradient background
    yy, xx = np.mgrid[0:H, 0:W]
    bg = (100 + 80 * np.sin(xx / 97.0) * np.cos(yy / 61.0)).astype(np.float32)
    x = torch.from_numpy((frame[0] * 20 + bg)[None]).cuda().contiguous(); p = torch.from_numpy((pat[0] * 60 + bg * 0.7)[None]).cuda().contiguous()
exact = te.xcorrvol_batch(x[None], p, D, 9, algo="exact")[0]
fast = te.xcorrvol_batch(x[None], p, D, 9, algo="fast")[0]
def win_stats(img, left, right):       # img [H,W] f64 -> mean, var at unclamped centres [-left, W+right)
    e = F.pad(img[None, None], (left + 4, right + 4, 4, 4), mode="replicate")
    k = torch.ones(1, 1, 9, 9, dtype=torch.float64, device=img.device)
    s1 = F.conv2d(e, k)[0, 0]; s2 = F.conv2d(e * e, k)[0, 0]
    mean = s1 / 81; var = (s2 - s1 * mean).clamp_min(0)
    return mean, var
xd, pd = x[0].double(), p[0].double()
ma, va = win_stats(xd, 0, 0); mb, vb = win_stats(pd, D - 1, 0)
ca = ma[H // 2, W // 2]; cb = mb[H // 2, W // 2 + D - 1]
Fa = 1 + 81 * (ma - ca) ** 2 / va.clamp_min(1e-30); Fbp = 1 + 81 * (mb - cb) ** 2 / vb.clamp_min(1e-30)
worst = 0; worst_tol = 0
hist = {}
for d in range(D):
    Fb = Fbp[:, D - 1 - d:D - 1 - d + W]
    err = (fast[d] - exact[d]).abs().double()
    tol = 1e-6 + 1e-5 * exact[d].abs().double()
    g = (Fa * Fb).sqrt()
    c = err / (6e-8 * g)
    worst = max(worst, c.max().item())
    r = err / tol
    worst_tol = max(worst_tol, r.max().item())
    for lo, hi in ((1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 1e9)):
        msk = (g >= lo) & (g < hi)
        if msk.any():
            a = hist.setdefault(lo, [0, 0.0, 0.0]); a[0] += int(msk.sum()); a[1] = max(a[1], r[msk].max().item()); a[2] = max(a[2], c[msk].max().item())
print(kind, H, W, D, "max err/(eps*sqrt(FaFb)) =", worst, " max err/tol =", worst_tol)
for lo in sorted(hist): print("  sqrt(FaFb) >= %-4g n=%-10d max err/tol %.3f  max c %.2f" % (lo, hist[lo][0], hist[lo][1], hist[lo][2]))
