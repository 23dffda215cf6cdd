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
    r = int(rs.choice([1, 2, 5, 5, 5, 7]))
    N = int(rs.randint(1, 5)); H = int(rs.randint(r + 1, 150)); W = int(rs.randint(r + 1, 300))
    x = torch.from_numpy((rs.rand(N, 1, H, W) * 3 + rs.randn(N, 1, 1, 1)).astype(np.float32)).cuda()
    y, s = te.lcn(x, r, 0.05)
    if r == 5:                                               # algo='fast' against the f64 kernel, the suite's tolerance
        yf, sf = te.lcn(x, r, 0.05, algo="fast")
        if not bool(((yf - y).abs() <= y.abs() * 1e-5 + 1e-6).all() and ((sf - s).abs() <= s.abs() * 1e-5 + 1e-6).all()):
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
