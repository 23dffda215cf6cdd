"""A7: RectifiedPatternSimilarityLoss (pattern warp + HIP photometric loss + masked mean) against values,
warped pattern and d loss / d disp captured from the reference module (tests/golden/pattern_loss.npz)."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, golden

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", ["census_sad", "mse"])
@pytest.mark.parametrize("use_std", [True, False])
def test_pattern_loss_vs_reference(name, use_std):
    from connecting_the_dots_amd import torchext as te
    g = golden("pattern_loss")
    H, W = g["im"].shape[2:]
    mod = te.RectifiedPatternSimilarityLoss(H, W, dev(g["pattern"]), loss_type=name, loss_eps=0.5, algo="exact")
    disp = dev(g["disp"]).requires_grad_(True)
    val, proj = mod(disp, dev(g["im"]), dev(g["std"]) if use_std else None)
    val.backward()
    tag = "%s_%d" % (name, use_std)
    # The warp is ATen's grid_sample on both sides, CPU there and GPU here: its bilinear weight ix - floor(ix)
    # cancels at |ix| ~ 40, i.e. ~4e-6 absolute, which the two implementations round differently.
    assert_close(proj.detach().cpu().numpy(), g["proj_" + tag], rtol=1e-5, atol=2e-5, what="proj " + tag)
    assert_close(val.item(), g["val_" + tag], rtol=1e-4, atol=0, what="val " + tag)
    # the gradient w.r.t. disparity goes through grid_sample's backward (ATen, atomics): tolerance, not bits
    assert_close(disp.grad.cpu().numpy(), g["gdisp_" + tag], rtol=2e-3, atol=2e-6, what="grad " + tag)


# ----------------------------------------------------------------------------------------------------------
# N1: fused kernels (algo='fast'): warp + block loss + masked mean, forward and backward
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["census_sad", "mse"])
@pytest.mark.parametrize("use_std", [True, False])
def test_fused_pattern_loss_vs_reference(name, use_std):
    from connecting_the_dots_amd import torchext as te
    g = golden("pattern_loss")
    H, W = g["im"].shape[2:]
    mod = te.RectifiedPatternSimilarityLoss(H, W, dev(g["pattern"]), loss_type=name, loss_eps=0.5, algo="fast")
    disp = dev(g["disp"]).requires_grad_(True)
    val, proj = mod(disp, dev(g["im"]), dev(g["std"]) if use_std else None)
    val.backward()
    tag = "%s_%d" % (name, use_std)
    assert_close(proj.detach().cpu().numpy(), g["proj_" + tag], rtol=1e-5, atol=2e-5, what="proj " + tag)
    assert_close(val.item(), g["val_" + tag], rtol=1e-4, atol=0, what="val " + tag)
    assert_close(disp.grad.cpu().numpy(), g["gdisp_" + tag], rtol=2e-3, atol=2e-6, what="grad " + tag)


def test_fused_matches_composition_at_training_size():
    """16 x 432 x 512, census_sad, eps 0.5 (the production call, networks.py:344,376): the fused kernels against
    the composition grid_sample + exact photometric kernels + masked mean, value / warped pattern / gradient"""
    from connecting_the_dots_amd import torchext as te
    torch.manual_seed(5)
    B, H, W = 16, 432, 512
    pattern = torch.randn(1, 3, H, W, device="cuda")
    im = torch.randn(B, 1, H, W, device="cuda")
    std = 0.05 + torch.rand(B, 1, H, W, device="cuda")
    disp0 = torch.rand(B, 1, H, W, device="cuda") * 60
    out = {}
    for algo in ("exact", "fast"):
        mod = te.RectifiedPatternSimilarityLoss(H, W, pattern, algo=algo)
        d = disp0.clone().requires_grad_(True)
        val, proj = mod(d, im, std)
        val.backward()
        out[algo] = (val.item(), proj.detach(), d.grad)
    assert abs(out["fast"][0] - out["exact"][0]) <= 1e-5 * abs(out["exact"][0])
    # bilinear weights ix - floor(ix) cancel at |ix| ~ 500 (ulp 3e-5) and multiply pattern differences of ~4
    assert float((out["fast"][1] - out["exact"][1]).abs().max()) <= 4e-4
    ge, gf = out["exact"][2], out["fast"][2]
    # ATen's grid_sample backward and the exact block loss on one side, one fused kernel on the other; the warp
    # gradient (p[x1] - p[x0]) amplifies the 1e-4 differences of the warped values
    # gradient (p[x1] - p[x0]) amplifies the 1e-4 differences of the warped values, and a census-SAD pair whose
    # diff lies within those 1e-4 of zero flips its sign between the two paths: a statistical bound
    bad = (gf - ge).abs() > 2e-3 * ge.abs() + 1e-3 * float(ge.abs().max())
    assert float(bad.float().mean()) < 1e-3, float(bad.float().mean())
    assert float((gf - ge).abs().mean()) <= 1e-3 * float(ge.abs().mean())


def test_fused_terms_support_cross_rank_ratio():
    """terms() must carry gradient through the numerator so that a ratio of all-reduced sums differentiates"""
    from connecting_the_dots_amd import torchext as te
    torch.manual_seed(6)
    B, H, W = 2, 40, 96
    pattern = torch.randn(1, 1, H, W, device="cuda")
    im, std = torch.randn(B, 1, H, W, device="cuda"), 0.1 + torch.rand(B, 1, H, W, device="cuda")
    mod = te.RectifiedPatternSimilarityLoss(H, W, pattern, algo="fast")
    d1 = (torch.rand(B, 1, H, W, device="cuda") * 20).requires_grad_(True)
    d2 = d1.detach().clone().requires_grad_(True)
    v, _ = mod(d1, im, std)
    v.backward()
    num, den, _ = mod.terms(d2, im, std)
    (num / den).backward()
    assert torch.allclose(d1.grad, d2.grad, rtol=1e-5, atol=1e-9)


def test_multi_level_launch_equals_per_level_calls():
    """N2: four pyramid levels (480x640 ... 60x80, as exp_synph.py trains on) in one launch each way give the
    very same numbers as four single-level calls (same tile code, same fixed-order reductions)"""
    from connecting_the_dots_amd import torchext as te
    torch.manual_seed(8)
    B = 2
    disps, ims, stds, pats = [], [], [], []
    for s in range(4):
        H, W = 480 >> s, 640 >> s
        pats.append(torch.randn(1, 1, H, W, device="cuda"))
        ims.append(torch.randn(B, 1, H, W, device="cuda"))
        stds.append(None if s == 3 else 0.05 + torch.rand(B, 1, H, W, device="cuda"))
        disps.append(torch.rand(B, 1, H, W, device="cuda") * (60 >> s))
    da = [d.clone().requires_grad_(True) for d in disps]
    vals, terms, projs = te.pattern_loss_multi(da, ims, stds, pats, "census_sad", 0.5)
    w = torch.tensor([1.0, 0.5, 0.25, 2.0], device="cuda")
    (vals * w).sum().backward()
    for s in range(4):
        d = disps[s].clone().requires_grad_(True)
        v, p, t = te.pattern_loss(d, ims[s], stds[s], pats[s], "census_sad", 0.5)
        (v * w[s]).backward()
        assert torch.equal(v, vals[s]) and torch.equal(t, terms[s]) and torch.equal(p, projs[s])
        assert torch.equal(d.grad, da[s].grad)
