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
# hard pixels as the scan kernel defines them (run past d_clamped counted once)
N, D, H, W = vol.shape
d = torch.arange(D, device=vol.device).view(1, D, 1, 1)
wcol = torch.arange(W, device=vol.device).view(1, 1, 1, W)
masked = torch.where(d > wcol + 4, torch.full_like(vol, float("-inf")), vol)
top2 = masked.topk(2, dim=1).values
hard = top2[:, 1] >= top2[:, 0] - 1e-5
print("hard pixels: %d of %d (%.4f%%); by column block of 64: %s" % (hard.sum().item(), hard.numel(), 100 * hard.float().mean().item(), hard.float().view(N, H, W // 64, 64).mean((0, 1, 3)).cpu().numpy().round(5)))
gap = (top2[:, 0] - top2[:, 1])
print("gap quantiles", torch.quantile(gap.flatten()[::7].float(), torch.tensor([0.001, 0.01, 0.1, 0.5], device=gap.device)).cpu().numpy())
print("exact zeros in vol: %.4f%%, |vol|<1e-6: %.4f%%" % (100 * (vol == 0).float().mean().item(), 100 * (vol.abs() < 1e-6).float().mean().item()))
tie = top2[:, 0] == top2[:, 1]
print("exact non-run ties:", tie.sum().item(), "positions (f,h,w):", tie.nonzero()[:10].cpu().numpy().tolist())
idx_f, best_f = te.xcorrvol_argmax(x, pl, 128, 9, algo="fast", rerank_eps=-1.0)
idx_r, best_r = te.xcorrvol_argmax(x, pl, 128, 9, algo="fast", rerank_eps=1e-5)
print("pixels changed by the re-rank:", (idx_f != idx_r).sum().item())
