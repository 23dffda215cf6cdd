"""Random soak of nn / crosscheck against a float64 brute force (distinct random points: no ties):
    python tools/fuzz_nn.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    n0, n1 = int(rs.randint(1, 3000)), int(rs.randint(1, 3000))
    dt = torch.float64 if rs.randint(0, 3) == 0 else torch.float32
    a = torch.from_numpy(rs.randn(n0, 3)).to(dt).cuda(); b = torch.from_numpy(rs.randn(n1, 3)).to(dt).cuda()
    i01 = te.nn(a, b); i10 = te.nn(b, a)
    d = ((a.double()[:, None, :] - b.double()[None, :, :]) ** 2).sum(-1)
    r01 = d.argmin(1); r10 = d.argmin(0)
    # (f32 distances can order two nearly equidistant candidates differently from f64: accept a candidate whose f64
    #  distance is within rounding of the best)
    ok01 = (i01 == r01) | ((d.gather(1, i01.clamp(min=0)[:, None])[:, 0] - d.min(1).values).abs() <= 1e-5 * d.min(1).values + 1e-7)
    ok10 = (i10 == r10) | ((d.t().gather(1, i10.clamp(min=0)[:, None])[:, 0] - d.min(0).values).abs() <= 1e-5 * d.min(0).values + 1e-7)
    cc = te.crosscheck(i01, i10)
    cr = (i10[i01.clamp(min=0)] == torch.arange(n0, device="cuda")).to(torch.uint8)
    if not bool(ok01.all()) or not bool(ok10.all()) or not torch.equal(cc, cr):
        bad += 1
        print("case %d n0=%d n1=%d %s: nn %d/%d off, crosscheck %d off" % (case, n0, n1, dt, int((~ok01).sum()), int((~ok10).sum()), int((cc != cr).sum())), flush=True)
print("fuzz_nn: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
