"""GPU parity of the photometric block loss (forward, deterministic gather backward) and of the SAD / census
cost volume, through the reference-shaped torchext API; bit-exact vs the reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from tests.util import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def reference_order_kernels(monkeypatch):
    """this module pins the reference-order kernels (bit-exact asserts): the package default is algo='fast'"""
    monkeypatch.setenv("CTD_NCC_ALGO", "exact")
    monkeypatch.setenv("CTD_PHOTO_ALGO", "exact")
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


# ----------------------------------------------------------------------------------------------------------
# algo='fast': tolerance-level kernels (free summation order, v_rsq_f32, pair-symmetric backward)
# ----------------------------------------------------------------------------------------------------------
def _grad_close(got, ref, go_scale, what):
    """|a-b| <= 1e-5|b| + 1e-6 * (scale of the gradient): a gradient entry is a signed sum of ~2*bs^2 terms of
    size <= go/bs^2, so its absolute rounding floor scales with go"""
    assert_close(got, ref, rtol=1e-5, atol=1e-6 * go_scale, what=what)


@pytest.mark.parametrize("ty", TYPES)
@pytest.mark.parametrize("shape,bs", [((2, 1, 37, 150), 9), ((1, 3, 20, 70), 5), ((1, 1, 9, 9), 9), ((2, 2, 64, 64), 3),
                                      ((1, 1, 5, 200), 7), ((1, 1, 1, 1), 9)])
def test_fast_matches_oracle(te, oracle, ty, shape, bs):
    """interior tiles, border tiles (tap multiplicities > 1), images smaller than the block, C > 1"""
    rs = np.random.RandomState(bs + sum(shape))
    es = rs.randn(*shape).astype(np.float32)
    ta = (es + 0.3 * rs.randn(*shape)).astype(np.float32)
    go = rs.rand(shape[0], 1, shape[2], shape[3]).astype(np.float32)
    eps = 0.5
    ref = oracle.photometric_fwd(es, ta, bs, TYPES.index(ty), eps)
    gref = oracle.photometric_bwd(es, ta, go, bs, TYPES.index(ty), eps)
    e = dev(es).requires_grad_(True)
    out = te.photometric_loss(e, dev(ta), bs, ty, eps, algo="fast")
    assert_close(out.detach().cpu().numpy(), ref, what="fwd %s" % ty)
    out.backward(dev(go))
    _grad_close(e.grad.cpu().numpy(), gref, 1.0, "bwd %s" % ty)


def test_fast_golden_and_sign_ties(te):
    """committed reference outputs; es == ta makes every census diff exactly 0 (sign 0, gradient 0) and a
    constant offset makes diff tiny but non-zero: the sign is settled in reference arithmetic"""
    g = golden("photometric")
    n = 0
    for k, (B, C, H, W, bs) in enumerate(g["cases"]):
        if int(bs) % 2 == 0:
            continue
        for ty in range(4):
            for eps in (0.1, 0.5):
                n += 1
                e = dev(g["es_%d" % k]).requires_grad_(True)
                out = te.photometric_loss(e, dev(g["ta_%d" % k]), int(bs), TYPES[ty], eps, algo="fast")
                assert_close(out.detach().cpu().numpy(), g["fwd_%d_%d_%g" % (k, ty, eps)], what="fwd %d %d" % (k, ty))
                out.backward(dev(g["go_%d" % k]))
                _grad_close(e.grad.cpu().numpy(), g["bwd_%d_%d_%g" % (k, ty, eps)],
                            float(np.abs(g["go_%d" % k]).max()), "bwd %d %d %g" % (k, ty, eps))
    assert n >= 8
    x = dev(np.random.RandomState(0).randn(1, 1, 24, 80).astype(np.float32)).requires_grad_(True)
    out = te.photometric_loss(x, x.detach().clone(), 9, "census_sad", 0.5, algo="fast")
    assert float(out.abs().max()) == 0.0
    out.backward(torch.ones_like(out))
    assert float(x.grad.abs().max()) == 0.0


def test_fast_full_size_properties(te):
    """BASELINE config-2 size (16 x 432 x 512): fast == exact within tolerance on every pixel, forward and backward"""
    torch.manual_seed(3)
    es = torch.randn(16, 1, 432, 512, device="cuda")
    ta = es + 0.2 * torch.randn_like(es)
    go = torch.rand(16, 1, 432, 512, device="cuda")
    for ty in ("census_sad", "sad"):
        a = es.clone().requires_grad_(True)
        b = es.clone().requires_grad_(True)
        fa = te.photometric_loss(a, ta, 9, ty, 0.5, algo="fast")
        fb = te.photometric_loss(b, ta, 9, ty, 0.5, algo="exact")
        assert bool(((fa - fb).abs() <= 1e-5 * fb.abs() + 1e-6).all())
        fa.backward(go)
        fb.backward(go)
        bad = (a.grad - b.grad).abs() > 1e-5 * b.grad.abs() + 1e-6
        assert int(bad.sum()) == 0, "%s: %d of %d gradient entries out of tolerance, max %.3g" % (
            ty, int(bad.sum()), bad.numel(), float((a.grad - b.grad).abs().max()))


@pytest.mark.parametrize("ty", TYPES)
@pytest.mark.parametrize("shape", [(24, 150, 40, 9), (37, 64, 7, 5), (9, 70, 33, 9), (16, 200, 70, 3)])
def test_costvol_fast_vs_oracle(te, oracle, ty, shape):
    """A6, algo='fast': image borders (both clamps), disparities beyond one LDS chunk (32), D not a multiple of 8"""
    H, W, D, bs = shape
    rs = np.random.RandomState(sum(shape))
    im = rs.randn(H, W).astype(np.float32)
    pat = rs.randn(H, W).astype(np.float32)
    ref = oracle.costvol(im, pat, D, bs, TYPES.index(ty), 0.5, nthreads=4)
    got = te.costvol(dev(im), dev(pat), D, bs, ty, 0.5, algo="fast").cpu().numpy()
    assert_close(got, ref, what="costvol %s %s" % (ty, (shape,)))
    # batched frames with per-frame patterns
    ims, pats = np.stack([im, pat]), np.stack([pat, im])
    gb = te.costvol(dev(ims), dev(pats), D, bs, ty, 0.5, algo="fast").cpu().numpy()
    assert_close(gb[0], ref, what="batched")


@pytest.mark.parametrize("ty", ["sad", "mse"])
@pytest.mark.parametrize("shape", [(24, 64, 40), (13, 256, 33), (30, 260, 128), (20, 512, 28), (9, 8, 5), (40, 300, 27),
                                   (5, 1028, 61), (12, 516, 1), (17, 128, 130)])
def test_costvol_separable_sad_mse_vs_oracle(te, oracle, ty, shape):
    """A6, SAD / MSE with block 9 and W % 4 == 0: the separable path (replicate-border box filter of |P[r][c-d] - I[r][c]|
    through the NCC volume kernel's pipeline + the recomputed last four columns).  Every output against the oracle's
    composition of photometric_loss_forward over shifted patterns: partial and exact column tiles, a tile plus four
    columns, bands shorter than the block, D = 1 / 27 / 28 (14 pairs) / 61 / 128 (split pair) / 130, both image borders."""
    H, W, D = shape
    rs = np.random.RandomState(sum(shape))
    im = rs.randn(H, W).astype(np.float32)
    pat = rs.randn(H, W).astype(np.float32)
    ref = oracle.costvol(im, pat, D, 9, TYPES.index(ty), 0.5, nthreads=4)
    got = te.costvol(dev(im), dev(pat), D, 9, ty, 0.5, algo="fast").cpu().numpy()
    assert_close(got, ref, what="separable costvol %s %s" % (ty, (shape,)))
    for cols in (slice(0, 8), slice(W - 8, W)):                   # the border columns on their own (both clamps)
        assert_close(got[..., cols], ref[..., cols], what="border columns")
    # batched frames with per-frame patterns, and a shared pattern
    ims, pats = np.stack([im, pat, im]), np.stack([pat, im, pat])
    gb = te.costvol(dev(ims), dev(pats), D, 9, ty, 0.5, algo="fast").cpu().numpy()
    assert_close(gb[0], ref, what="batched, per-frame patterns")
    assert_close(gb[2], ref, what="batched, per-frame patterns (frame 2)")
    gs = te.costvol(dev(ims), dev(pat), D, 9, ty, 0.5, algo="fast").cpu().numpy()
    assert_close(gs[0], ref, what="batched, shared pattern")
    assert np.array_equal(gs[0], gs[2])
