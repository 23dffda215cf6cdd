import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from connecting_the_dots_amd import torchext as te
import bench
frames, pattern = bench.make_inputs(4, 0, torch.device("cuda"))
pl, _ = te.lcn(pattern, 5, 0.05); pl = pl[0].contiguous()
x, _ = te.lcn(frames, 5, 0.05)
vol = te.xcorrvol_batch(x, pl, 128, 9, algo="fast")
m = vol.max(1, keepdim=True).values
c = (vol >= m - 1e-5).sum(1)
print("candidates per pixel: mean %.3f, frac>1: %.4f, max %d" % (c.float().mean().item(), (c > 1).float().mean().item(), c.max().item()))
print("by column (first 16 cols): ", c[0, :, :16].float().mean(0).cpu().numpy().round(1))
print("pattern lcn: min/max", pl.min().item(), pl.max().item(), " nan:", torch.isnan(vol).sum().item())
for name, fn in [("argmax_rerank", lambda: te.xcorrvol_argmax(x, pl, 128, 9, algo="fast")), ("torch.argmax", lambda: vol.argmax(1))]:
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 5 * 1e3, "ms for 4 frames (incl. volume for the first)")
