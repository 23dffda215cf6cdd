"""GPU parity of the photometric block loss (forward, deterministic gather backward) and of the SAD / census
cost volume, through the reference-shaped torchext API; bit-exact vs the reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, golden

pytestmark = pytest.mark.gpu
TYPES = ["mse", "sad", "census_mse", "census_sad"]


@pytest.fixture(scope="module")
def te():
    from connecting_the_dots_amd import torchext
    return torchext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("ty", range(4))
def test_forward_backward_golden_bit_exact(te, ty):
    g = golden("photometric")
    for k, (B, C, H, W, bs) in enumerate(g["cases"]):
        for eps in (0.1, 0.5):
            es = dev(g["es_%d" % k]).requires_grad_(True)
            ta = dev(g["ta_%d" % k])
            out = te.photometric_loss(es, ta, int(bs), TYPES[ty], eps)
            assert np.array_equal(out.detach().cpu().numpy(), g["fwd_%d_%d_%g" % (k, ty, eps)]), (k, ty, eps)
            out.backward(dev(g["go_%d" % k]))
            assert np.array_equal(es.grad.cpu().numpy(), g["bwd_%d_%d_%g" % (k, ty, eps)]), (k, ty, eps)
            assert ta.grad is None


@pytest.mark.parametrize("shape", [(1, 1, 5, 7, 9), (2, 2, 33, 70, 9), (1, 1, 3, 3, 9), (1, 3, 9, 130, 5), (1, 1, 1, 40, 3),
                                   (1, 1, 40, 1, 9), (1, 1, 12, 12, 4)])
@pytest.mark.parametrize("ty", range(4))
def test_vs_oracle_ragged(te, oracle, shape, ty):
    B, C, H, W, bs = shape
    rs = np.random.RandomState(sum(shape) + ty)
    es = rs.randn(B, C, H, W).astype(np.float32)
    ta = rs.randn(B, C, H, W).astype(np.float32)
    go = rs.randn(B, 1, H, W).astype(np.float32)
    e = dev(es).requires_grad_(True)
    out = te.photometric_loss(e, dev(ta), bs, TYPES[ty], 0.5)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.photometric_fwd(es, ta, bs, ty, 0.5, nthreads=4))
    out.backward(dev(go))
    assert np.array_equal(e.grad.cpu().numpy(), oracle.photometric_bwd(es, ta, go, bs, ty, 0.5))


def test_training_shape_census_sad_vs_oracle(te, oracle):
    """The only production call: block 9, census_sad, eps 0.5 (networks.py:344,376), one 480x640 frame."""
    rs = np.random.RandomState(77)
    es = rs.randn(1, 1, 480, 640).astype(np.float32)
    ta = rs.randn(1, 1, 480, 640).astype(np.float32)
    go = rs.rand(1, 1, 480, 640).astype(np.float32)
    e = dev(es).requires_grad_(True)
    out = te.photometric_loss(e, dev(ta), 9, "census_sad", 0.5)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.photometric_fwd(es, ta, 9, 3, 0.5, nthreads=8))
    out.backward(dev(go))
    assert np.array_equal(e.grad.cpu().numpy(), oracle.photometric_bwd(es, ta, go, 9, 3, 0.5))


def test_matches_pytorch_formulation_and_autograd(te):
    """The reference's own cross-check (functions.py:120-147): kernel vs unfold formulation, forward and grad."""
    rs = np.random.RandomState(1)
    es0 = rs.rand(2, 2, 20, 24).astype(np.float64)
    ta = dev(rs.rand(2, 2, 20, 24).astype(np.float64))
    go = dev(rs.randn(2, 1, 20, 24))
    for name in TYPES:
        a = dev(es0).requires_grad_(True)
        b = dev(es0).requires_grad_(True)
        o1 = te.photometric_loss(a, ta, 9, name, 0.5)
        o2 = te.photometric_loss_pytorch(b, ta, 9, name, 0.5)
        assert_close(o1.detach().cpu().numpy(), o2.detach().cpu().numpy(), rtol=1e-10, atol=1e-12, what=name)
        o1.backward(go)
        o2.backward(go)
        assert_close(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-9, atol=1e-11, what=name + " grad")


def test_backward_is_deterministic(te):
    rs = np.random.RandomState(2)
    es = dev(rs.randn(2, 1, 64, 80).astype(np.float32))
    ta = dev(rs.randn(2, 1, 64, 80).astype(np.float32))
    go = dev(rs.randn(2, 1, 64, 80).astype(np.float32))
    grads = []
    for _ in range(3):
        e = es.clone().requires_grad_(True)
        te.photometric_loss(e, ta, 9, "census_sad", 0.5).backward(go)
        grads.append(e.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


def test_type_strings_and_errors(te):
    x = torch.rand(1, 1, 8, 8).cuda()
    assert torch.equal(te.photometric_loss(x, x, 3, "SAD"), te.photometric_loss(x, x, 3, "sad"))   # case-insensitive
    with pytest.raises(Exception, match="invalid loss type"):
        te.photometric_loss(x, x, 3, "ncc")
    with pytest.raises(RuntimeError):
        te.photometric_loss(x.cpu(), x.cpu(), 3)
    with pytest.raises(RuntimeError):
        te.photometric_loss(x, x[:, :, ::2], 3)


@pytest.mark.parametrize("ty", range(4))
def test_costvol_golden_bit_exact(te, ty):
    g = golden("costvol")
    vol = te.costvol(dev(g["im"]), dev(g["pat"]), int(g["D"]), int(g["bs"]), TYPES[ty], 0.5)
    assert np.array_equal(vol.cpu().numpy(), g["vol_%d" % ty])
    assert np.array_equal(vol.argmin(0).cpu().numpy(), g["argmin_%d" % ty])


def test_costvol_batch_vs_oracle(te, oracle):
    rs = np.random.RandomState(9)
    im = rs.rand(2, 17, 90).astype(np.float32)
    pat = rs.rand(2, 17, 90).astype(np.float32)
    vol = te.costvol(dev(im), dev(pat), 20, 9, "census_sad", 0.5).cpu().numpy()
    for f in range(2):
        assert np.array_equal(vol[f], oracle.costvol(im[f], pat[f], 20, 9, 3, 0.5, nthreads=4))
