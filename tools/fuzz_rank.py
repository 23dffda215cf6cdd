"""Random-shape soak of the ranked fast argmax against the reference-order kernel (not part of the test suite):
    python tools/fuzz_rank.py [cases] [seed] [big]
Every case: indices of the ranked call with and without a volume == the exact kernel's, volume == the plain fast volume
and within tolerance of the exact one, best scores within tolerance, prepared == unprepared."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te
from tests import workloads

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = len(sys.argv) > 3
bad = 0
t0 = time.time()
for case in range(cases):
    N = int(rs.randint(1, 5)); H = int(rs.randint(1, 90)); W = 4 * int(rs.randint(1, 150)); D = int(rs.randint(1, 301))
    if big:
        N = int(rs.randint(1, 18)); H = int(rs.randint(40, 300)); W = 4 * int(rs.randint(60, 280)); D = int(rs.randint(20, 400))
    kind = rs.randint(0, 4)
    if kind == 2 and (H < 7 or W < 8):
        kind = 1                                                   # (LCN's reflection padding needs radius < size)
    if kind == 0:                                                  # white noise
        a = rs.randn(N, 1, H, W).astype(np.float32); b = rs.randn(1, H, W).astype(np.float32)
    elif kind == 1:                                                # noise with flat patches (listed windows / runs)
        a = rs.randn(N, 1, H, W).astype(np.float32); b = rs.randn(1, H, W).astype(np.float32)
        for _ in range(3):
            y, x = rs.randint(0, H), rs.randint(0, W)
            b[:, y:y + 12, x:x + 14] = rs.randn() * 0.3
            a[rs.randint(0, N), :, y:y + 10, x:x + 11] = rs.randn()
        b[:, : H // 2, :10] = 0.125
    elif kind == 2:                                                # LCN'd frames against the LCN'd dot pattern
        fr = np.stack([workloads.uniform_frame(int(rs.randint(1 << 20)), H, W).reshape(H, W) for _ in range(N)])[:, None]
        pat = workloads.syn_dot_pattern(H, W, seed=int(rs.randint(1 << 20))).reshape(1, 1, H, W)
        a = te.lcn(torch.from_numpy(fr).cuda(), 5, 0.05)[0].cpu().numpy()
        b = te.lcn(torch.from_numpy(pat).cuda(), 5, 0.05)[0][0].cpu().numpy().reshape(1, H, W)
    else:                                                          # smooth + offset (cancellation) and a periodic pattern
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        b = (np.sin(xx / 3.0) + 0.1 * rs.randn(H, W)).astype(np.float32)[None] + 5.0
        a = (np.sin((xx[None] + rs.randint(0, 9)) / 3.0) + 0.1 * rs.randn(N, H, W)).astype(np.float32)[:, None] + 5.0
    A = torch.from_numpy(np.ascontiguousarray(a)).cuda(); B = torch.from_numpy(np.ascontiguousarray(b)).cuda()
    idx_e, best_e, vol_e = te.xcorrvol_argmax(A, B, D, 9, return_volume=True, algo="exact")
    idx_v, best_v, vol_v = te.xcorrvol_argmax(A, B, D, 9, return_volume=True, algo="fast")
    idx_n, best_n = te.xcorrvol_argmax(A, B, D, 9, algo="fast")
    plain = te.xcorrvol_batch(A, B, D, 9, algo="fast")
    h = te.prepare_pattern(B, N, D, 9)
    idx_p, best_p, vol_p = te.xcorrvol_argmax(A, B, D, 9, return_volume=True, algo="fast", prepared=h)
    idx_q, _ = te.xcorrvol_argmax(A, B, D, 9, algo="fast", prepared=h)
    tol = vol_e.abs() * 1e-5 + 1e-6
    tolb = vol_e.abs().amax(1) * 1e-5 + 2e-6
    problems = []
    if not torch.equal(idx_v, idx_e): problems.append("idx(volume) %d" % int((idx_v != idx_e).sum()))
    if not torch.equal(idx_n, idx_e): problems.append("idx(no volume) %d" % int((idx_n != idx_e).sum()))
    if not torch.equal(vol_v, plain): problems.append("ranked volume != plain volume")
    if not bool(((vol_v - vol_e).abs() <= tol).all()): problems.append("volume tolerance %g" % float(((vol_v - vol_e).abs() - tol).max()))
    if not bool(((best_v - best_e).abs() <= tolb).all()): problems.append("best (volume) tolerance %g" % float(((best_v - best_e).abs() - tolb).max()))
    if not bool(((best_n - best_e).abs() <= tolb).all()): problems.append("best (no volume) tolerance %g" % float(((best_n - best_e).abs() - tolb).max()))
    if not (torch.equal(idx_p, idx_v) and torch.equal(best_p, best_v) and torch.equal(vol_p, vol_v) and torch.equal(idx_q, idx_n)):
        problems.append("prepared != unprepared")
    if problems:
        bad += 1
        print("case %d N=%d H=%d W=%d D=%d kind=%d: %s" % (case, N, H, W, D, kind, "; ".join(problems)), flush=True)
    if case % 1000 == 999:
        print("... %d cases, %d bad, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("fuzz: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
