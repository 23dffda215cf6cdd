"""Random-shape soak of the separable SAD / MSE cost volume (block 9, W % 4 == 0: the all-D pipeline + border rule) against
the reference-order kernel:   python tools/fuzz_costvol_sep.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, t0 = 0, time.time()
for case in range(cases):
    N = int(rs.randint(1, 6)); H = int(rs.randint(1, 120)); W = 4 * int(rs.randint(2, 200)); D = int(rs.randint(1, 260))
    if rs.randint(0, 4) == 0:
        W = 256 * int(rs.randint(1, 4))                          # exact column tiles: the loader's right-border rule
    ty = ["sad", "mse"][int(rs.randint(0, 2))]
    per_frame = bool(rs.randint(0, 2))
    im = torch.from_numpy((rs.randn(N, H, W) * rs.choice([1.0, 0.1, 10.0])).astype(np.float32) + np.float32(rs.choice([0.0, 3.0]))).cuda()
    pat = torch.from_numpy(rs.randn(*((N, H, W) if per_frame else (H, W))).astype(np.float32)).cuda()
    ce = te.costvol(im, pat, D, 9, ty, 0.5, algo="exact")
    cf = te.costvol(im, pat, D, 9, ty, 0.5, algo="fast")
    x = float(((cf - ce).abs() - (ce.abs() * 1e-5 + 1e-6)).max())
    if x > 0 or not bool(torch.isfinite(cf).all()):
        bad += 1
        print("case %d N=%d H=%d W=%d D=%d %s per_frame=%d: excess %g" % (case, N, H, W, D, ty, per_frame, x), flush=True)
    if case % 100 == 99:
        print("... %d cases, %d bad, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("fuzz_costvol_sep: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
