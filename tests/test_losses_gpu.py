"""A8-A10: fused DispToDepth, Sobel+DisparityLoss and geometric loss kernels (forward and backward) against
values and autograd gradients captured from the reference modules (tests/golden/losses.npz).  The reference's
arithmetic here is ATen's (conv / bmm / grid_sample orders unspecified): parity by tolerance."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def te():
    from connecting_the_dots_amd import torchext
    return torchext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_disp_to_depth_fwd_bwd(te):
    g = golden("losses")
    d = dev(g["d2d_disp"]).requires_grad_(True)
    depth = te.DispToDepth(567.6, 0.075)(d)
    assert_close(depth.detach().cpu().numpy(), g["d2d_depth"], rtol=1e-6, atol=0, what="depth")
    depth.backward(dev(g["d2d_go"]))
    assert_close(d.grad.cpu().numpy(), g["d2d_grad"], rtol=2e-6, atol=0, what="grad")
    assert (d.grad[d <= 0] == 0).all()                         # relu gate (networks.py:319)


def test_disparity_loss_with_edge(te):
    g = golden("losses")
    d = dev(g["dl_disp"]).requires_grad_(True)
    e = dev(g["dl_edge"]).requires_grad_(True)
    val = te.DisparityLoss()(d, e)
    assert_close(val.item(), g["dl_val"], rtol=1e-5, atol=0, what="value")
    val.backward()
    assert_close(d.grad.cpu().numpy(), g["dl_gdisp"], rtol=1e-4, atol=1e-8, what="grad disp")
    assert_close(e.grad.cpu().numpy(), g["dl_gedge"], rtol=1e-4, atol=1e-9, what="grad edge")


def test_disparity_loss_without_edge(te):
    g = golden("losses")
    d = dev(g["dl_disp"]).requires_grad_(True)
    val = te.disparity_loss(d)
    assert_close(val.item(), g["dl_noedge_val"], rtol=1e-5, atol=0, what="value")
    (val * 3.0).backward()                                     # non-unit upstream gradient
    assert_close(d.grad.cpu().numpy(), 3.0 * g["dl_noedge_gdisp"], rtol=1e-4, atol=1e-8, what="grad disp")


def test_disparity_loss_matches_torch_ops_on_gpu(te):
    """Same formula with stock torch ops on the GPU (conv2d / exp / log / mean) incl. borders of a ragged image."""
    rs = np.random.RandomState(4)
    disp = dev((np.cumsum(rs.rand(3, 1, 37, 70), 3) * 3).astype(np.float32)).requires_grad_(True)
    edge = dev(rs.rand(3, 1, 37, 70).astype(np.float32)).requires_grad_(True)
    kx = torch.tensor([[-5, -4, 0, 4, 5], [-8, -10, 0, 10, 8], [-10, -20, 0, 20, 10], [-8, -10, 0, 10, 8],
                       [-5, -4, 0, 4, 5]], dtype=torch.float64).div(240).float().cuda()
    x = torch.nn.functional.pad(disp, (2, 2, 2, 2), mode="replicate")
    gx = torch.nn.functional.conv2d(x, kx.view(1, 1, 5, 5))
    gy = torch.nn.functional.conv2d(x, kx.t().contiguous().view(1, 1, 5, 5))
    gm = torch.sqrt(gx ** 2 + gy ** 2 + 1e-8)
    b0, b1 = 0.0503428816795, 1.07274045944
    pdf = (1 - edge) / b0 * torch.exp(-gm / b0) + edge / b1 * torch.exp(-gm / b1)
    ref = torch.mean(-torch.log(pdf.clamp(min=1e-4)))
    gd, ge = torch.autograd.grad(ref, (disp, edge))
    val = te.disparity_loss(disp, edge)
    gd2, ge2 = torch.autograd.grad(val, (disp, edge))
    assert_close(val.item(), ref.item(), rtol=1e-5, atol=0, what="value")
    assert_close(gd2.cpu().numpy(), gd.cpu().numpy(), rtol=2e-4, atol=1e-8, what="grad disp")
    assert_close(ge2.cpu().numpy(), ge.cpu().numpy(), rtol=2e-4, atol=1e-9, what="grad edge")


@pytest.mark.parametrize("tag,clamp", [("c", 0.1), ("nc", -1.0)])
def test_geometric_loss(te, tag, clamp):
    g = golden("losses")
    H, W = g["ge_depth0"].shape[2:]
    mod = te.ProjectionDepthSimilarityLoss(torch.from_numpy(g["ge_K"]), torch.from_numpy(g["ge_Ki"]), H, W, clamp=clamp)
    assert_close(mod.ray.numpy(), g["ge_ray"][0], rtol=1e-6, atol=1e-7, what="ray")
    d0 = dev(g["ge_depth0"]).requires_grad_(True)
    d1 = dev(g["ge_depth1"]).requires_grad_(True)
    val = mod(d0, d1, dev(g["ge_R0"]), dev(g["ge_t0"]), dev(g["ge_R1"]), dev(g["ge_t1"]))
    assert_close(val.item(), g["ge_%s_val" % tag], rtol=2e-5, atol=0, what="value")
    val.backward()
    # the sampled-depth gradient is amplified by the pixel-scale bilinear slopes; ATen CPU vs this kernel differ
    # in the rounding of ix - floor(ix)
    assert_close(d0.grad.cpu().numpy(), g["ge_%s_g0" % tag], rtol=2e-3, atol=2e-7, what="grad depth0")
    assert_close(d1.grad.cpu().numpy(), g["ge_%s_g1" % tag], rtol=2e-3, atol=2e-7, what="grad depth1")


def test_idx_to_depth_matches_disp_to_depth(te):
    """additive op: depth straight from the int64 argmax indices (+ offset) == DispToDepth on the float disparity"""
    idx = torch.randint(0, 128, (3, 40, 64), device="cuda", dtype=torch.int64)
    got = te.idx_to_depth(idx, 567.6 * 0.075, 1.0)
    ref = te.DispToDepth(567.6, 0.075)(idx.to(torch.float32) + 1.0)
    assert got.dtype == torch.float32 and torch.equal(got, ref)
    with pytest.raises(RuntimeError):
        te.idx_to_depth(idx.to(torch.int32), 1.0)


@pytest.mark.parametrize("B,H,W,clamp", [(1, 5, 7, -1.0), (3, 33, 130, 0.1), (15, 432, 512, 0.1)])
def test_geometric_symmetric_launch_equals_the_two_calls(te, B, H, W, clamp):
    """ctd_geometric_sym_fwd_f32 (both directions in one launch, means by the last workgroup): the same bits as
    ctd_geometric_fwd_f32 called twice (accumulate 0 / 1, views swapped), call after call on the same ticket"""
    from connecting_the_dots_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(B * 100 + H)
    d0 = dev(1.0 + rs.rand(B, 1, H, W).astype(np.float32))
    d1 = dev(1.0 + rs.rand(B, 1, H, W).astype(np.float32))
    K = np.array([[0.9 * W, 0, W / 2.0], [0, 0.9 * W, H / 2.0], [0, 0, 1]], np.float32)
    mod = te.ProjectionDepthSimilarityLoss(torch.from_numpy(K), torch.from_numpy(np.linalg.inv(K.astype(np.float64)).astype(np.float32)), H, W, clamp=clamp)
    ray, Kd = mod.ray.cuda().contiguous(), dev(K)
    R0 = dev(np.stack([np.eye(3, dtype=np.float32)] * B)); R1 = R0.clone()
    R1[:, 0, 1] = 0.01; R1[:, 1, 0] = -0.01
    t0 = dev(rs.randn(B, 3).astype(np.float32) * 0.02); t1 = dev(rs.randn(B, 3).astype(np.float32) * 0.02)
    s = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(L.ctd_geometric_workspace_bytes(2 * B, H, W), dtype=torch.uint8, device="cuda")
    two = torch.empty((), device="cuda")
    p = lambda t: t.data_ptr()
    assert L.ctd_geometric_fwd_f32(p(d0), p(d1), p(ray), p(Kd), p(R0), p(t0), p(R1), p(t1), p(two), 0, B, H, W, clamp, p(ws), ws.numel(), 0, s) == 0
    assert L.ctd_geometric_fwd_f32(p(d1), p(d0), p(ray), p(Kd), p(R1), p(t1), p(R0), p(t0), p(two), 1, B, H, W, clamp, p(ws), ws.numel(), 0, s) == 0
    ticket = torch.zeros(65, dtype=torch.int32, device="cuda")
    for rep in range(3):
        ws.fill_(0xAB)                                   # stale partials of another call must not matter
        one = torch.full((), float("nan"), device="cuda")
        assert L.ctd_geometric_sym_fwd_f32(p(d0), p(d1), p(ray), p(Kd), p(R0), p(t0), p(R1), p(t1), p(one), B, H, W, clamp,
                                           p(ws), ws.numel(), p(ticket), 0, s) == 0
        assert torch.equal(one, two), (rep, one.item(), two.item())
        assert int(ticket.abs().sum().item()) == 0
    assert torch.equal(mod(d0, d1, R0, t0, R1, t1), two)           # the module takes the one-launch path
    assert L.ctd_geometric_sym_fwd_f32(p(d0), p(d1), p(ray), p(Kd), p(R0), p(t0), p(R1), p(t1), p(two), B, H, W, clamp,
                                       p(ws), 16, p(ticket), 0, s) != 0                  # workspace too small
