"""N3: nn / crosscheck / proj_nn HIP kernels (integer results) -- bit-exact against the goldens captured from
the reference's ext_cpu (tests/golden/nn_ops.npz), against the oracle on fresh seeds, and through properties at
sizes the O(n^2) oracle would not finish."""
import numpy as np
import pytest
import torch

from tests import workloads
from tests.util import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def te():
    from connecting_the_dots_amd import torchext
    return torchext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag,dt", [("f32", np.float32), ("f64", np.float64)])
def test_golden(te, tag, dt):
    g = golden("nn_ops")
    a, b = workloads.nn_case(3, dt)
    i01, i10 = te.nn(dev(a), dev(b)), te.nn(dev(b), dev(a))
    assert i01.dtype == torch.int64 and i01.shape == (a.shape[0],)
    assert np.array_equal(i01.cpu().numpy(), g["nn01_" + tag])
    assert np.array_equal(i10.cpu().numpy(), g["nn10_" + tag])
    cc = te.crosscheck(i01, i10)
    assert cc.dtype == torch.uint8
    assert np.array_equal(cc.cpu().numpy(), g["cc_" + tag])
    xyz0, xyz1, K = workloads.proj_case(4, dt)
    for ps in (1, 3, 4, 5):
        out = te.proj_nn(dev(xyz0), dev(xyz1), dev(K), ps)
        assert np.array_equal(out.cpu().numpy(), g["proj%d_%s" % (ps, tag)]), ps


@pytest.mark.parametrize("n0,n1", [(1, 1), (257, 1023), (1000, 1024), (513, 2500)])
def test_nn_vs_oracle_tile_edges(te, oracle, n0, n1):
    """candidate tiles of 1024: sizes around the tile edge, duplicated points across tiles (lowest index wins)"""
    rs = np.random.RandomState(n0 + n1)
    a = rs.normal(size=(n0, 3)).astype(np.float32)
    b = rs.normal(size=(n1, 3)).astype(np.float32)
    if n1 > 1100:
        b[1050] = b[3]
        a[0] = b[3]
    out = te.nn(dev(a), dev(b)).cpu().numpy()
    assert np.array_equal(out, oracle.nn(a, b))
    if n1 > 1100:
        assert out[0] == 3


def test_nn_empty_and_errors(te):
    a = torch.zeros((4, 3), device="cuda")
    assert te.nn(a, torch.zeros((0, 3), device="cuda")).tolist() == [-1] * 4      # ext.h:30: min_arg stays -1
    assert te.nn(torch.zeros((0, 3), device="cuda"), a).shape == (0,)
    with pytest.raises(RuntimeError):
        te.nn(a, torch.zeros((4, 2), device="cuda"))
    with pytest.raises(RuntimeError):
        te.nn(a.cpu(), a.cpu())
    with pytest.raises(RuntimeError):
        te.crosscheck(a, a)                                                       # not int64
    out = te.crosscheck(torch.tensor([0, 5, -1, 1], device="cuda"), torch.tensor([0, 3], device="cuda"))
    assert out.tolist() == [1, 0, 0, 1]                                           # out-of-range and negative -> 0
    assert te.nn(a, a).requires_grad is False


def test_nn_full_frame_properties(te):
    """config-scale point clouds (2 x 60x80 depth maps): self-match is the identity, matching a permuted copy
    recovers the permutation, and the cross-check of mutual matches is all ones"""
    rs = np.random.RandomState(5)
    n = 9600
    a = rs.normal(size=(n, 3)).astype(np.float32)
    perm = rs.permutation(n)
    A, B = dev(a), dev(a[perm])
    assert torch.equal(te.nn(A, A), torch.arange(n, device="cuda"))
    i_ab = te.nn(A, B)                       # a[i] == b[j]  <=>  perm[j] == i
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    assert np.array_equal(i_ab.cpu().numpy(), inv)
    i_ba = te.nn(B, A)
    assert np.array_equal(i_ba.cpu().numpy(), perm)
    assert te.crosscheck(i_ab, i_ba).all()


def test_proj_nn_vs_oracle_and_identity(te, oracle):
    xyz0, xyz1, K = workloads.proj_case(11, np.float32, B=3, H=60, W=80)
    for ps in (3, 7):
        out = te.proj_nn(dev(xyz0), dev(xyz1), dev(K), ps).cpu().numpy()
        assert np.array_equal(out, oracle.proj_nn(xyz0, xyz1, K, ps))
    # a cloud matched against itself with a 1x1 patch returns each pixel's own flat index
    B, H, W = 2, 24, 32
    _, xyz1, K = workloads.proj_case(12, np.float64, B=B, H=H, W=W)
    out = te.proj_nn(dev(xyz1), dev(xyz1), dev(K), 1).cpu().numpy()
    own = np.arange(B * H * W).reshape(B, H, W)
    assert (out == own).mean() > 0.99 and ((out == own) | (out == -1) | (np.abs(out - own) <= W + 1)).all()
