import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from connecting_the_dots_amd import torchext as te
torch.manual_seed(0)
H, W, D = 40, 512, 16
a = torch.randn(1, 1, H, W, device="cuda"); b = torch.randn(1, H, W, device="cuda")
ref = te.xcorrvol_batch(a, b, D, 9, algo="exact")[0]
fast = te.xcorrvol_batch(a, b, D, 9, algo="fast")[0]
err = (fast - ref)
print("max |err| cols 0-3:", err[:, :, 0:4].abs().amax((1,)).cpu().numpy().round(3))
print("max |err| cols 256-259:", err[:, :, 256:260].abs().amax((1,)).cpu().numpy().round(3))
print("d=0 row 20 cols 0..7 fast", fast[0, 20, :8].cpu().numpy().round(4), "ref", ref[0, 20, :8].cpu().numpy().round(4))
print("d=5 row 20 cols 254..261 fast", fast[5, 20, 254:262].cpu().numpy().round(4), "ref", ref[5, 20, 254:262].cpu().numpy().round(4))
