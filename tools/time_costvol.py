"""Times the SAD / census cost volumes (A6) at BASELINE config 1 / 2 (512x432x128, per frame) and config 4 (1024x1024x256):
    python tools/time_costvol.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te
from tests import workloads
for (H, W, D, N) in ((432, 512, 128, 4), (1024, 1024, 256, 1)):
    fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
    pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)).cuda()
    fr = fr.reshape(N, 1, H, W)
    x, _ = te.lcn(fr.contiguous(), 5, 0.05)
    p, _ = te.lcn(pat.reshape(1, 1, H, W).contiguous(), 5, 0.05)
    x, p = x[:, 0].contiguous(), p[0, 0].contiguous()
    for kind in ("sad", "mse", "census_sad", "census_mse"):
        for _ in range(30):
            v = te.costvol(x, p, D, 9, kind, 0.5, algo="fast")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            v = te.costvol(x, p, D, 9, kind, 0.5, algo="fast")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("%dx%dx%d x %d frames  %-11s %.4f ms per call  %.4f ms per frame  %.0f Mpix*disp/s  %.2f TB/s of volume writes" % (
            W, H, D, N, kind, dt * 1e3, dt * 1e3 / N, N * H * W * D / dt / 1e6, N * H * W * D * 4 / dt / 1e12), flush=True)
