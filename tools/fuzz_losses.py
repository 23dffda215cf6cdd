"""Random-shape soak of DispToDepth, the edge-aware disparity loss and the symmetric two-view geometric loss (value and
gradients) against stock-PyTorch formulations of model/networks.py:395-412, 436-503:   python tools/fuzz_losses.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import torchext as te

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, t_start = 0, time.time()

def torch_disparity_loss(disp, edge):
    kx = torch.tensor([[-5, -4, 0, 4, 5], [-8, -10, 0, 10, 8], [-10, -20, 0, 20, 10], [-8, -10, 0, 10, 8],
                       [-5, -4, 0, 4, 5]], dtype=torch.float64).div(240).float().to(disp.device)     # networks.py:395-404
    x = F.pad(disp, (2, 2, 2, 2), mode="replicate")
    gx = F.conv2d(x, kx.view(1, 1, 5, 5)); gy = F.conv2d(x, kx.t().contiguous().view(1, 1, 5, 5))
    gm = torch.sqrt(gx ** 2 + gy ** 2 + 1e-8)
    b0, b1 = 0.0503428816795, 1.07274045944
    pdf = (1 - edge) / b0 * torch.exp(-gm / b0) + edge / b1 * torch.exp(-gm / b1)
    return torch.mean(-torch.log(pdf.clamp(min=1e-4)))

def torch_geo_dir(d0, d1, K, ray, R0, t0, R1, t1, clamp, H, W):
    B = d0.shape[0]
    xyz = d0.reshape(B, -1, 1) * ray.unsqueeze(0)
    xyz = torch.bmm(xyz - t0.reshape(B, 1, 3), R0)
    xyz = torch.bmm(xyz, R1.transpose(1, 2)) + t1.reshape(B, 1, 3)
    uvd = xyz @ K.T
    d = uvd[:, :, 2:3]
    uv = uvd[:, :, :2] / (F.relu(d) + 1e-12)
    grid = torch.stack((2 * (uv[:, :, 0] / (W - 1) - 0.5), 2 * (uv[:, :, 1] / (H - 1) - 0.5)), dim=2).view(B, H, W, 2)
    d10 = F.grid_sample(d1, grid, padding_mode="border", align_corners=False)
    diff = torch.abs(d.view(B, 1, H, W) - d10)
    return (torch.clamp(diff, 0, clamp) if clamp > 0 else diff).mean()

def rel(a, b, scale=None):
    s = float(b.abs().max()) if scale is None else scale
    return float((a - b).abs().max()) / max(s, 1e-30)

for case in range(cases):
    B = int(rs.randint(1, 5)); H = int(rs.randint(6, 70)); W = int(rs.randint(6, 200))
    problems = []
    disp = torch.from_numpy((np.cumsum(rs.rand(B, 1, H, W), 3) * 3 + rs.rand(B, 1, H, W) * float(rs.choice([0, 0.2, 2]))).astype(np.float32)).cuda().requires_grad_(True)
    edge = torch.from_numpy(rs.rand(B, 1, H, W).astype(np.float32)).cuda().requires_grad_(True)
    ref = torch_disparity_loss(disp, edge); gr = torch.autograd.grad(ref, (disp, edge))
    val = te.disparity_loss(disp, edge); gv = torch.autograd.grad(val, (disp, edge))
    if abs(val.item() - ref.item()) > 2e-5 * abs(ref.item()): problems.append("disparity loss value")
    if rel(gv[0], gr[0]) > 5e-4 or rel(gv[1], gr[1]) > 5e-4: problems.append("disparity loss grads %g %g" % (rel(gv[0], gr[0]), rel(gv[1], gr[1])))
    bf = 567.6 * 0.075
    dep = te.DispToDepth(567.6, 0.075)(disp.detach())
    if rel(dep, bf / disp.detach()) > 1e-6: problems.append("disp->depth")
    K = torch.tensor([[0.9 * W, 0, W / 2.0], [0, 0.9 * W, H / 2.0], [0, 0, 1]], device="cuda")
    Ki = torch.linalg.inv(K.double()).float()
    clamp = float(rs.choice([-1.0, 0.1, 0.5]))
    mod = te.ProjectionDepthSimilarityLoss(K, Ki, H, W, clamp=clamp)
    d0 = torch.from_numpy((1.0 + rs.rand(B, 1, H, W)).astype(np.float32)).cuda().requires_grad_(True)
    d1 = torch.from_numpy((1.0 + rs.rand(B, 1, H, W)).astype(np.float32)).cuda().requires_grad_(True)
    ax = rs.randn(B, 3) * 0.02
    Rs = []
    for a in ax:
        th = np.linalg.norm(a); k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rs.append(np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx)
    R1 = torch.from_numpy(np.stack(Rs).astype(np.float32)).cuda(); R0 = torch.eye(3, device="cuda").repeat(B, 1, 1)
    t0 = torch.from_numpy((rs.randn(B, 3) * 0.02).astype(np.float32)).cuda(); t1 = torch.from_numpy((rs.randn(B, 3) * 0.02).astype(np.float32)).cuda()
    ray = mod.ray.cuda()
    ref = torch_geo_dir(d0, d1, K, ray, R0, t0, R1, t1, clamp, H, W) + torch_geo_dir(d1, d0, K, ray, R1, t1, R0, t0, clamp, H, W)
    g0r, g1r = torch.autograd.grad(ref, (d0, d1))
    val = mod(d0, d1, R0, t0, R1, t1); g0, g1 = torch.autograd.grad(val, (d0, d1))
    if abs(val.item() - ref.item()) > 5e-5 * abs(ref.item()) + 1e-7: problems.append("geometric value %g vs %g" % (val.item(), ref.item()))
    # the sampled-depth gradient jumps where a sampling position crosses a pixel boundary or |diff| crosses the clamp: a
    # rounding-level difference in the position flips single elements (ATen's CPU and GPU kernels differ the same way), so
    # count the elements that are off instead of taking the maximum
    sc = float(max(g0r.abs().max(), g1r.abs().max()))
    off = int(((g0 - g0r).abs() > 2e-3 * sc).sum() + ((g1 - g1r).abs() > 2e-3 * sc).sum())
    if off > max(12, 5e-4 * g0.numel()): problems.append("geometric grads: %d of %d elements off" % (off, 2 * g0.numel()))
    if problems:
        bad += 1
        print("case %d B=%d H=%d W=%d clamp=%g: %s" % (case, B, H, W, clamp, "; ".join(problems)), flush=True)
    if case % 200 == 199:
        print("... %d cases, %d bad, %.0f s" % (case + 1, bad, time.time() - t_start), flush=True)
print("fuzz_losses: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
