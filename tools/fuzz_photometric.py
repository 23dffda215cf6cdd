"""Random-shape soak of the tolerance-level photometric kernels and cost volumes against the reference-order ones
(all four loss types, forward + backward, random eps / block size):   python tools/fuzz_photometric.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
types = ["mse", "sad", "census_mse", "census_sad"]
bad, t0 = 0, time.time()

def excess(a, b, rel=1e-5, ab=1e-6):
    return float(((a - b).abs() - (b.abs() * rel + ab)).max())

for case in range(cases):
    N = int(rs.randint(1, 4)); H = int(rs.randint(1, 60)); W = int(rs.randint(1, 300)); D = int(rs.randint(1, 140))
    bs = int(rs.choice([3, 5, 9, 9])); ty = types[int(rs.randint(0, 4))]; eps = float(rs.choice([0.1, 0.5, 1e-3]))
    es = torch.from_numpy(rs.randn(N, 1, H, W).astype(np.float32)).cuda().requires_grad_(True)
    ta = torch.from_numpy(rs.randn(N, 1, H, W).astype(np.float32)).cuda()
    go = torch.from_numpy(rs.rand(N, 1, H, W).astype(np.float32)).cuda()
    problems = []
    le = te.photometric_loss(es, ta, bs, ty, eps, algo="exact")
    ge, = torch.autograd.grad(le, es, go)
    lf = te.photometric_loss(es, ta, bs, ty, eps, algo="fast")
    gf, = torch.autograd.grad(lf, es, go)
    x = excess(lf.detach(), le.detach())
    if x > 0: problems.append("fwd excess %g" % x)
    x = excess(gf, ge, 1e-5, 1e-6)
    if x > 0: problems.append("bwd excess %g" % x)
    im = ta[:, 0].contiguous(); pat = es.detach()[0, 0].contiguous()
    ce = te.costvol(im, pat, D, bs, ty, eps, algo="exact")
    cf = te.costvol(im, pat, D, bs, ty, eps, algo="fast")
    x = excess(cf, ce)
    if x > 0 or not bool(torch.isfinite(cf).all()): problems.append("costvol excess %g" % x)
    if problems:
        bad += 1
        print("case %d N=%d H=%d W=%d D=%d bs=%d %s eps=%g: %s" % (case, N, H, W, D, bs, ty, eps, "; ".join(problems)), flush=True)
    if case % 500 == 499:
        print("... %d cases, %d bad, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("fuzz_photometric: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
