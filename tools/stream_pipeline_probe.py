"""Config-2 steps (LCN -> prepared ranked NCC volume + argmax) issued round-robin on S streams: consecutive batches overlap
their latency-bound side kernels (LCN / pre-pass of batch k+1 beside fix-up / tail of batch k; the all-D kernel owns the
whole chip).  ms per step for S = 1, 2, 3.    python tools/stream_pipeline_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads
from connecting_the_dots_amd import torchext as te
H, W, D, N = 432, 512, 128, 16
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
p, _ = te.lcn(pat, 5, 0.05)
p = p[0].contiguous()
torch.cuda.synchronize()
for S in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(S)]
    prepared = []
    for s in streams:
        with torch.cuda.stream(s):
            prepared.append(te.prepare_pattern(p, N, D, 9))
    torch.cuda.synchronize()
    held = [None] * S

    def step(k):
        s = streams[k % S]
        with torch.cuda.stream(s):
            x, _ = te.lcn(fr, 5, 0.05)
            held[k % S] = (x,) + te.xcorrvol_argmax(x, p, D, 9, return_volume=True, prepared=prepared[k % S])
    for k in range(600):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(60):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 60
    # same indices on every stream
    ok = all(torch.equal(held[0][1], h[1]) for h in held)
    print("%d stream(s): %.4f ms per step  (%.0f Mpix*disp/s)  indices equal across streams: %s" % (S, dt * 1e3, N * H * W * D / dt / 1e6, ok), flush=True)
    del held, prepared
