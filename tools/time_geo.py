"""Profiling target: disparity -> depth + the symmetric geometric loss of config 3's 8 frame pairs, 200 calls
(rocprofv3 --kernel-trace --stats -- python tools/time_geo.py [variant.so])"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from connecting_the_dots_amd import torchext as te
H, W, B = 432, 512, 8
rs = np.random.RandomState(0)
idx = torch.from_numpy(rs.randint(0, 128, (2 * B, H, W))).cuda()
K = torch.tensor([[567.6, 0, 324.7], [0, 570.2, 250.1], [0, 0, 1]], device="cuda")
geo = te.ProjectionDepthSimilarityLoss(K, torch.linalg.inv(K.double()).float(), H, W, clamp=0.1)
R0 = torch.eye(3, device="cuda").repeat(B, 1, 1); R1 = R0.clone(); R1[:, 0, 2] = 0.01; R1[:, 2, 0] = -0.01
t0 = torch.from_numpy(rs.randn(B, 3).astype(np.float32) * 0.02).cuda(); t1 = torch.from_numpy(rs.randn(B, 3).astype(np.float32) * 0.02).cuda()
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for it in range(200):
    depth = te.idx_to_depth(idx, 567.6 * 0.075, 1.0).view(-1, 1, H, W)
    loss = geo(depth[:B], depth[B:], R0, t0, R1, t1)
    if it % 4 == 0:
        big.fill_(1)                 # the step's other kernels push these operands out of the caches
torch.cuda.synchronize()
print("loss", float(loss))
