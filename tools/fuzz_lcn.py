"""Random-shape soak of the LCN kernel against a float64 PyTorch formulation of LCN.tforward (model/networks.py:523-533):
    python tools/fuzz_lcn.py [cases] [seed]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    r = int(rs.choice([1, 2, 3, 4, 5, 5, 5, 6, 7]))
    N = int(rs.randint(1, 5)); H = int(rs.randint(r + 1, 150)); W = int(rs.randint(r + 1, 300))
    kind = case % 4
    if kind == 0:                                            # uniform with a DC level per frame
        xn = rs.rand(N, 1, H, W) * 3 + rs.randn(N, 1, 1, 1)
    elif kind == 1:                                          # structured light: flat zero background, sparse bright samples
        xn = (rs.rand(N, 1, H, W) < 0.06) * (0.5 + 0.5 * rs.rand(N, 1, H, W))
    elif kind == 2:                                          # the same on a low-noise dark level, a bright sample on a regular grid
        xn = (rs.rand(N, 1, H, W) < 0.06) * 0.9 + 0.02 + 0.01 * rs.rand(N, 1, H, W)
        xn[:, :, 8::16, 32::64] = 0.9
    else:                                                    # flat regions at different levels (one f32 order's var is as
        xn = np.round(rs.rand(N, 1, H, W) * 2) * 0.25 + 0.003 * rs.randn(N, 1, H, W)   # good as another's only up to E[x^2] 2^-24)
    x = torch.from_numpy(xn.astype(np.float32)).cuda()
    y, s = te.lcn(x, r, 0.05)
    if kind != 3:                                 # algo='fast' against the f64 kernel, the suite's tolerance
        yf, sf = te.lcn(x, r, 0.05, algo="fast")
        # (kind 2 -- a low-noise level with windows that hold no bright sample -- sits on the variance floor of 1e-6: an f32
        # one-pass variance is good to E[(x - c)^2] * 2^-24 * (roundings) there, 3e-6 absolute of std at these levels)
        atol = 3e-6 if kind == 2 else 1e-6
        # (windows of 9 .. 81 samples have, now and then, a variance far below their mean square: there the exact kernel's
        # own f32 tail E[x^2] - avg^2 is as noisy as any other order's -- 5e-5 relative below radius 5)
        rtol = 1e-5 if r >= 5 else 5e-5
        if not bool(((yf - y).abs() <= y.abs() * rtol + atol).all() and ((sf - s).abs() <= s.abs() * rtol + atol).all()):
            bad += 1
            print("case %d N=%d H=%d W=%d: fast vs exact y %g std %g" % (case, N, H, W, float((yf - y).abs().max()), float((sf - s).abs().max())), flush=True)
    xd = x.double(); n = float((2 * r + 1) ** 2)
    p = F.pad(xd, (r, r, r, r), mode="reflect")
    k = torch.ones(1, 1, 2 * r + 1, 2 * r + 1, dtype=torch.float64, device="cuda")
    avg = F.conv2d(p, k) / n; sq = F.conv2d(F.pad((x * x).double(), (r, r, r, r), mode="reflect"), k) / n
    sd = torch.sqrt((sq - avg * avg).clamp(min=0) + 1e-6) + 0.05
    yr = (xd - avg) / sd
    ey = float((y.double() - yr).abs().max()); es = float((s.double() - sd).abs().max())
    # (the reference's own f32 E[x^2] - avg^2 loses ~1e-6 absolute of the variance: compare at 2e-4 of the outputs' scale)
    if ey > 2e-4 * float(yr.abs().max()) + 1e-5 or es > 2e-4 * float(sd.abs().max()):
        bad += 1
        print("case %d N=%d H=%d W=%d r=%d: y %g std %g" % (case, N, H, W, r, ey, es), flush=True)
print("fuzz_lcn: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
