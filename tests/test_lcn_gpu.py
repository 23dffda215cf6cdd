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
                                   (1, 7, 300, 0)])
def test_lcn_bit_exact_vs_oracle(te, oracle, shape):
    N, H, W, r = shape
    rs = np.random.RandomState(N * H + W)
    x = (rs.rand(N, 1, H, W) * 3 + rs.randn(N, 1, 1, 1)).astype(np.float32)
    y0, s0 = oracle.lcn(x, r, 0.05)
    y, s = te.lcn(dev(x), r, 0.05)
    assert np.array_equal(s.cpu().numpy(), s0)
    assert np.array_equal(y.cpu().numpy(), y0)


def test_lcn_errors(te):
    with pytest.raises(RuntimeError):
        te.lcn(torch.rand(1, 1, 8, 8), 2, 0.05)            # CPU tensor
    with pytest.raises(RuntimeError):
        te.lcn(torch.rand(1, 1, 4, 8).cuda(), 4, 0.05)     # radius >= H
    with pytest.raises(RuntimeError):
        te.lcn(torch.rand(1, 2, 8, 8).cuda(), 2, 0.05)     # C != 1
