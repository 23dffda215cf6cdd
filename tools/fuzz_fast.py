"""Randomised shapes: fast NCC volume vs reference-order volume (tolerance), re-ranked argmax vs fused exact argmax."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from connecting_the_dots_amd import torchext as te
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    C = int(rs.choice([1, 1, 1, 2, 3])); N = int(rs.randint(1, 4)); bs = int(rs.choice([9, 9, 9, 7, 5, 3]))
    H = int(rs.randint(1, 120)); W = int(rs.choice([rs.randint(1, 70), 4 * rs.randint(1, 140), rs.randint(60, 600)])); D = int(rs.choice([1, 2, rs.randint(1, 40), rs.randint(40, 300)]))
    per_frame = bool(rs.rand() < 0.3)
    kind = rs.choice(["normal", "flat", "offset", "dots"])
    a = rs.randn(N, C, H, W).astype(np.float32); b = rs.randn(N if per_frame else 1, C, H, W).astype(np.float32)
    if kind == "flat": a[:, :, : H // 2] = 0.5; b[:, :, :, : max(1, W // 3)] = -1.0
    if kind == "offset": a = a * 5 + 100; b = b * 9 + 40
    if kind == "dots": b = (rs.rand(*b.shape) < 0.1).astype(np.float32)
    A = torch.from_numpy(a).cuda(); B = torch.from_numpy(b if per_frame else b[0]).cuda()
    try:
        ex = te.xcorrvol_batch(A, B, D, bs, algo="exact"); fa = te.xcorrvol_batch(A, B, D, bs, algo="fast")
    except RuntimeError as e:
        print("cfg", (N, C, H, W, D, bs, per_frame, kind), "raised", str(e)[:80]); continue
    err = (fa - ex).abs(); bad = err > ex.abs() * 1e-5 + 1e-6
    ok = not bool(bad.any()) and not bool(torch.isnan(fa).any())
    if C == 1:
        i1, _ = te.xcorrvol_argmax(A, B, D, bs, algo="fast"); i2, _ = te.xcorrvol_argmax(A, B, D, bs, algo="exact")
        ok = ok and torch.equal(i1, i2)
    if not ok:
        n_bad += 1
        print("FAIL", (N, C, H, W, D, bs, per_frame, kind), "bad", int(bad.sum()), "max err", float(err.max()))
print("done, failures:", n_bad)
