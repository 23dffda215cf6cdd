"""Randomised shapes: fast photometric fwd/bwd and cost volume vs the reference-order kernels."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from connecting_the_dots_amd import torchext as te
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
TYPES = ["mse", "sad", "census_mse", "census_sad"]
n_bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    B = int(rs.randint(1, 4)); C = int(rs.choice([1, 1, 2, 3])); bs = int(rs.choice([9, 9, 7, 5, 3]))
    H = int(rs.randint(1, 100)); W = int(rs.choice([rs.randint(1, 70), rs.randint(60, 300)])); ty = TYPES[rs.randint(4)]
    eps = float(rs.choice([0.1, 0.5]))
    es = torch.from_numpy(rs.randn(B, C, H, W).astype(np.float32)).cuda()
    ta = es + torch.from_numpy((rs.randn(B, C, H, W) * rs.choice([0.01, 0.3, 2.0])).astype(np.float32)).cuda()
    go = torch.from_numpy(rs.rand(B, 1, H, W).astype(np.float32)).cuda()
    out = {}
    for algo in ("exact", "fast"):
        e = es.clone().requires_grad_(True)
        o = te.photometric_loss(e, ta, bs, ty, eps, algo=algo); o.backward(go)
        out[algo] = (o.detach(), e.grad)
    f_ok = bool(((out["fast"][0] - out["exact"][0]).abs() <= 1e-5 * out["exact"][0].abs() + 1e-6).all())
    gb = (out["fast"][1] - out["exact"][1]).abs() > 1e-5 * out["exact"][1].abs() + 1e-6
    ok = f_ok and int(gb.sum()) == 0
    if C == 1 and B == 1 and H >= 2:
        D = int(rs.randint(1, 50))
        c1 = te.costvol(es[0, 0], ta[0, 0], D, bs, ty, eps, algo="exact"); c2 = te.costvol(es[0, 0], ta[0, 0], D, bs, ty, eps, algo="fast")
        ok = ok and bool(((c2 - c1).abs() <= 1e-5 * c1.abs() + 1e-6).all())
    if not ok:
        n_bad += 1; print("FAIL", (B, C, H, W, bs, ty, eps), "fwd ok", f_ok, "bad grads", int(gb.sum()), float((out["fast"][1] - out["exact"][1]).abs().max()))
print("done, failures:", n_bad)
