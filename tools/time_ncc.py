import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time, torch, numpy as np
from connecting_the_dots_amd import torchext as te
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, D = 432, 512, 128
a = torch.rand(N, 1, H, W, device="cuda"); b = torch.rand(1, H, W, device="cuda")
a, _ = te.lcn(a, 5, 0.05); b, _ = te.lcn(b[None], 5, 0.05); b = b[0].contiguous()
ref = None
for algo in sys.argv[2:] or ["exact", "fast"]:
    for _ in range(2): v = te.xcorrvol_batch(a, b, D, 9, algo=algo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 10
    e0.record()
    for _ in range(K): v = te.xcorrvol_batch(a, b, D, 9, algo=algo)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("%s N=%d: %.3f ms/step, %.1f us/frame, %.3e Mpix.disp/s, %.1f GB/s algorithmic" % (algo, N, ms, ms*1e3/N, N*H*W*D/ms/1e3, N*H*W*D*4.0625/ms/1e6))
    if ref is None: ref = v
    else:
        err = (v - ref).abs(); tol = 1e-5 * ref.abs() + 1e-6
        print("   vs %s: max abs err %.3e, outside tol %d / %d; argmax mismatches %d" % (sys.argv[2] if len(sys.argv)>2 else "exact", err.max().item(), (err > tol).sum().item(), err.numel(), (v.argmax(1) != ref.argmax(1)).sum().item()))
