"""Config 5: the own training step (connecting_the_dots_amd/train.py) -- loss parity for a fixed seed
between the HIP loss path (fused fast kernels / exact kernels) and a PyTorch-only loss path built from the
reference's own formulations (photometric_loss_pytorch, grid_sample, conv Sobel), and the step actually learns."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import workloads

pytestmark = pytest.mark.gpu

H, W, D, B = 48, 96, 24, 2


def batch(seed):
    rs = np.random.RandomState(seed)
    pat = workloads.syn_dot_pattern(H, W, seed=42)
    ir = np.stack([workloads.synth_ir(pat, rs, D, block=(12, 16))[0] for _ in range(B)])[:, None].astype(np.float32)
    return torch.from_numpy(pat[None, None]).cuda(), torch.from_numpy(ir).cuda()


def torch_only_loss(te, disp, edge, im, std, pattern, dp_weight=0.02):
    """networks.py:358-378 + :380-412 / :537-565 with stock PyTorch ops only"""
    Bn = disp.shape[0]
    u = torch.arange(W, dtype=torch.float32, device=disp.device).view(1, 1, -1).expand(1, H, -1)
    v = torch.arange(H, dtype=torch.float32, device=disp.device).view(1, -1, 1).expand(1, -1, W)
    u1 = u - disp.view(Bn, H, W)
    grid = torch.stack((2 * (u1 / (W - 1) - 0.5), (2 * (v / (H - 1) - 0.5)).expand(Bn, -1, -1)), dim=3)
    proj = F.grid_sample(pattern.mean(1, keepdim=True).expand(Bn, -1, -1, -1), grid, padding_mode="border",
                         align_corners=False)
    diff = te.photometric_loss_pytorch(proj, im, 9, "census_sad", 0.5)
    photo = (std * diff).sum() / std.sum()
    kx = torch.tensor([[-5, -4, 0, 4, 5], [-8, -10, 0, 10, 8], [-10, -20, 0, 20, 10], [-8, -10, 0, 10, 8],
                       [-5, -4, 0, 4, 5]], dtype=torch.float32, device=disp.device) / 240.0
    dp = F.pad(disp, (2, 2, 2, 2), mode="replicate")
    gx, gy = F.conv2d(dp, kx.view(1, 1, 5, 5)), F.conv2d(dp, kx.t().contiguous().view(1, 1, 5, 5))
    g = torch.sqrt(gx * gx + gy * gy + 1e-8)
    e = 1 - torch.sigmoid(edge)
    b0, b1 = 0.0503428816795, 1.07274045944
    pdf = (1 - e) / b0 * torch.exp(-g / b0) + e / b1 * torch.exp(-g / b1)
    return photo, dp_weight * (-torch.log(pdf.clamp(min=1e-4))).mean()


@pytest.mark.parametrize("algo", ["fast", "exact"])
def test_loss_parity_with_torch_only_path(algo):
    from connecting_the_dots_amd import torchext as te
    from connecting_the_dots_amd.train import DisparityTrainer, SmallDispEdgeNet
    pattern, ir = batch(1)
    pat_lcn, _ = te.lcn(pattern, 5, 0.05)
    torch.manual_seed(0)
    tr = DisparityTrainer(SmallDispEdgeNet(max_disp=D), pat_lcn, H, W, algo=algo)
    im, std = tr.lcn(ir)
    disp, edge = tr.net(im)
    vals = tr.loss_forward(disp, edge, im, std)
    ref = torch_only_loss(te, disp, edge, im, std, pat_lcn)
    for a, b, tol in zip(vals, ref, (2e-4, 1e-5)):      # the warp differs by ATen-vs-own bilinear rounding (1e-5 abs)
        assert abs(float(a) - float(b)) <= tol * abs(float(b)), (float(a), float(b))
    # gradients w.r.t. the network output agree as well
    ga = torch.autograd.grad(sum(vals), disp, retain_graph=True)[0]
    gb = torch.autograd.grad(sum(ref), disp)[0]
    assert float((ga - gb).abs().mean()) <= 2e-2 * float(gb.abs().mean())


def test_training_reduces_the_loss_and_times_every_bucket():
    from connecting_the_dots_amd import torchext as te
    from connecting_the_dots_amd.train import DisparityTrainer, SmallDispEdgeNet
    pattern, ir = batch(2)
    pat_lcn, _ = te.lcn(pattern, 5, 0.05)
    torch.manual_seed(0)
    tr = DisparityTrainer(SmallDispEdgeNet(max_disp=D), pat_lcn, H, W, lr=3e-3)
    first = sum(tr.train_step(ir))
    for _ in range(40):
        last = sum(tr.train_step(ir))
    assert np.isfinite(last) and last < first
    assert set(tr.watch.mean_ms()) == {"total", "forward", "loss", "backward", "optimizer"}
