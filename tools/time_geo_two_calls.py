import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import _lib, torchext as te
L = _lib.lib()
H, W, B = 432, 512, 8
rs = np.random.RandomState(0)
idx = torch.from_numpy(rs.randint(0, 128, (2 * B, H, W))).cuda()
K = torch.tensor([[567.6, 0, 324.7], [0, 570.2, 250.1], [0, 0, 1]], device="cuda")
geo = te.ProjectionDepthSimilarityLoss(K, torch.linalg.inv(K.double()).float(), H, W, clamp=0.1)
ray = geo.ray.cuda().contiguous()
R0 = torch.eye(3, device="cuda").repeat(B, 1, 1); R1 = R0.clone(); R1[:, 0, 2] = 0.01; R1[:, 2, 0] = -0.01
t0 = torch.from_numpy(rs.randn(B, 3).astype(np.float32) * 0.02).cuda(); t1 = torch.from_numpy(rs.randn(B, 3).astype(np.float32) * 0.02).cuda()
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
ws = torch.empty(L.ctd_geometric_workspace_bytes(2 * B, H, W), dtype=torch.uint8, device="cuda")
loss = torch.empty((), device="cuda"); s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
for it in range(200):
    depth = te.idx_to_depth(idx, 567.6 * 0.075, 1.0).view(-1, 1, H, W)
    d0, d1 = depth[:B], depth[B:]
    L.ctd_geometric_fwd_f32(p(d0), p(d1), p(ray), p(K), p(R0), p(t0), p(R1), p(t1), p(loss), 0, B, H, W, 0.1, p(ws), ws.numel(), 0, s)
    L.ctd_geometric_fwd_f32(p(d1), p(d0), p(ray), p(K), p(R1), p(t1), p(R0), p(t0), p(loss), 1, B, H, W, 0.1, p(ws), ws.numel(), 0, s)
    if it % 4 == 0:
        big.fill_(1)
torch.cuda.synchronize()
