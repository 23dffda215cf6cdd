"""GPU parity of the fused LCN kernel: bit-exact vs the oracle restatement, tolerance vs the
vectors captured from the reference's networks.LCN."""
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


def test_lcn_vs_reference_golden(te):
    g = golden("lcn_networks")
    for k in range(2):
        y, s = te.lcn(dev(g["x_%d" % k]), 5, 0.05)
        assert_close(s.cpu().numpy(), g["std_%d" % k], what="std %d" % k)
        assert_close(y.cpu().numpy(), g["y_%d" % k], rtol=2e-5, atol=2e-6, what="lcn %d" % k)
    y, s = te.LCN(2, 0.1)(dev(g["x_r2"]))
    assert_close(s.cpu().numpy(), g["std_r2"], what="std r2")
    assert_close(y.cpu().numpy(), g["y_r2"], rtol=2e-5, atol=2e-6, what="lcn r2")


@pytest.mark.parametrize("shape", [(2, 24, 32, 5), (1, 40, 53, 5), (3, 17, 130, 2), (1, 432, 512, 5), (1, 12, 13, 11),
                                   (1, 7, 300, 0), (2, 33, 65, 5), (1, 64, 96, 5), (1, 31, 31, 5), (1, 97, 34, 3)])
def test_lcn_bit_exact_vs_oracle(te, oracle, shape):
    N, H, W, r = shape
    rs = np.random.RandomState(N * H + W)
    x = (rs.rand(N, 1, H, W) * 3 + rs.randn(N, 1, 1, 1)).astype(np.float32)
    y0, s0 = oracle.lcn(x, r, 0.05)
    y, s = te.lcn(dev(x), r, 0.05)
    assert np.array_equal(s.cpu().numpy(), s0)
    assert np.array_equal(y.cpu().numpy(), y0)


@pytest.mark.parametrize("shape", [(2, 24, 32), (1, 40, 53), (1, 432, 512), (2, 33, 65), (1, 31, 31), (3, 97, 34), (1, 11, 300)])
def test_lcn_fast_within_tolerance(te, oracle, shape):
    """algo='fast' (f32 sliding box sums, radius 5): every output within 1e-5 |b| + 1e-6 of the oracle, on uniform frames,
    frames with a per-frame DC offset, binary patterns, structured-light frames (flat or low-noise dark background with
    sparse bright samples, one of them at every tile's centre -- the case the round-4 centring failed on) and the
    reference's own goldens (the contract for the LCN is a tolerance: ATen's conv2d summation order is unspecified)"""
    N, H, W = shape
    rs = np.random.RandomState(N * H + W)
    dots = ((rs.rand(N, 1, H, W) < 0.06) * (0.6 + 0.4 * rs.rand(N, 1, H, W))).astype(np.float32)
    dots_c = dots.copy()
    dots_c[:, :, 8::16, 32::64] = 0.9                      # a bright sample at the centre of every 64 x 16 tile
    for kind, x in enumerate(((rs.rand(N, 1, H, W) * 3 + rs.randn(N, 1, 1, 1)).astype(np.float32), rs.rand(N, 1, H, W).astype(np.float32),
                              (rs.rand(N, 1, H, W) < 0.1).astype(np.float32), dots, dots_c,
                              (dots_c + 0.02 + 0.01 * rs.rand(N, 1, H, W)).astype(np.float32))):
        y0, s0 = oracle.lcn(x, 5, 0.05)
        y, s = te.lcn(dev(x), 5, 0.05, algo="fast")
        assert_close(s.cpu().numpy(), s0, what="fast std, input kind %d" % kind)
        assert_close(y.cpu().numpy(), y0, what="fast lcn, input kind %d" % kind)


def test_lcn_fast_vs_reference_golden(te):
    g = golden("lcn_networks")
    for k in range(2):
        y, s = te.lcn(dev(g["x_%d" % k]), 5, 0.05, algo="fast")
        assert_close(s.cpu().numpy(), g["std_%d" % k], what="std %d" % k)
        assert_close(y.cpu().numpy(), g["y_%d" % k], rtol=2e-5, atol=2e-6, what="lcn %d" % k)
    # the other radii the f32 kernel serves (1 .. 7)
    y, s = te.LCN(2, 0.1, algo="fast")(dev(g["x_r2"]))
    assert_close(s.cpu().numpy(), g["std_r2"], what="std r2")
    assert_close(y.cpu().numpy(), g["y_r2"], rtol=2e-5, atol=2e-6, what="lcn r2")


@pytest.mark.parametrize("radius", [1, 3, 4, 6, 7, 9])
def test_lcn_fast_other_radii(te, oracle, radius):
    """algo='fast' for the radii 1 .. 7 (same kernel, tap loops of another length) against the oracle; 9 runs the f64 kernel"""
    rs = np.random.RandomState(radius)
    x = (rs.rand(2, 1, 45, 150) * 3 + rs.randn(2, 1, 1, 1)).astype(np.float32)
    y0, s0 = oracle.lcn(x, radius, 0.05)
    y, s = te.lcn(dev(x), radius, 0.05, algo="fast")
    assert_close(s.cpu().numpy(), s0, what="fast std r%d" % radius)
    assert_close(y.cpu().numpy(), y0, what="fast lcn r%d" % radius)


def test_lcn_errors(te):
    with pytest.raises(RuntimeError):
        te.lcn(torch.rand(1, 1, 8, 8), 2, 0.05)            # CPU tensor
    with pytest.raises(RuntimeError):
        te.lcn(torch.rand(1, 1, 4, 8).cuda(), 4, 0.05)     # radius >= H
    with pytest.raises(RuntimeError):
        te.lcn(torch.rand(1, 2, 8, 8).cuda(), 2, 0.05)     # C != 1


def test_lcn_datagen_variant_bit_exact(oracle):
    """data/lcn/lcn.pyx:16-58 (`lcn.normalize`): golden from the cythonized reference, plus the oracle on a batch of
    tile-straddling sizes; the zero border of width kernel_size is part of the contract"""
    from connecting_the_dots_amd import torchext as te
    g = golden("lcn_datagen")
    img = torch.from_numpy(g["img"]).cuda()
    for ks, eps in ((5, 0.05), (2, 0.1)):                  # the parameters the goldens were generated with
        y, s = te.lcn_normalize(img, ks, eps)
        ry, rs = oracle.lcn_datagen(g["img"], ks, eps)
        assert np.array_equal(y.cpu().numpy(), ry) and np.array_equal(s.cpu().numpy(), rs)
        assert np.array_equal(y.cpu().numpy(), g["y_%d" % ks]) and np.array_equal(s.cpu().numpy(), g["std_%d" % ks])
        assert float(y[:ks].abs().max()) == 0 and float(y[:, :ks].abs().max()) == 0 and float(s[-ks:].abs().max()) == 0
    rs_ = np.random.RandomState(4)
    batch = rs_.rand(3, 37, 150).astype(np.float32)
    y, s = te.lcn_normalize(torch.from_numpy(batch).cuda(), 5, 0.1)
    for k in range(3):
        ry, rstd = oracle.lcn_datagen(batch[k], 5, 0.1)
        assert np.array_equal(y[k].cpu().numpy(), ry) and np.array_equal(s[k].cpu().numpy(), rstd)
