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
    mod = te.RectifiedPatternSimilarityLoss(H, W, dev(g["pattern"]), loss_type=name, loss_eps=0.5)
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
