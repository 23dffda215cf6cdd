"""GPU parity of the fused call `lcn_xcorrvol_argmax` (ctd_lcn_xcorrvol_argmax_f32; streaming LCN + window statistics,
lcn_stream.hip -> all-D kernel -> fix-up -> tail): the LCN outputs against the CPU oracle (`exact`: bit for bit; `fast`:
tolerance on well-conditioned input), the indices against torch.argmax of the reference-order volume of the LCN output,
the materialised volume within the fast path's tolerance of that volume."""
import numpy as np
import pytest
import torch

from tests import workloads
from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def te():
    from connecting_the_dots_amd import torchext
    return torchext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def frames_of(kind, rs, N, H, W):
    if kind == "uniform":
        return rs.rand(N, 1, H, W).astype(np.float32)
    if kind == "offset":                                   # DC level per frame, larger than the deviation
        return (rs.rand(N, 1, H, W) * 3 + 10 * rs.randn(N, 1, 1, 1)).astype(np.float32)
    if kind == "dots":                                     # structured-light frame: dark flat background, sparse bright dots
        return ((rs.rand(N, 1, H, W) < 0.06) * (0.6 + 0.4 * rs.rand(N, 1, H, W))).astype(np.float32)
    if kind == "dots_noise":                               # the same on a noisy dark level
        return ((rs.rand(N, 1, H, W) < 0.06) * 0.9 + 0.02 + 0.01 * rs.rand(N, 1, H, W)).astype(np.float32)
    raise ValueError(kind)


def check(te, oracle, x, pat_lcn, D, lcn_algo, prepared=None, what=""):
    N, _, H, W = x.shape
    xd = dev(x)
    y, s, idx, best, vol = te.lcn_xcorrvol_argmax(xd, pat_lcn, D, 9, 5, 0.05, return_volume=True, lcn_algo=lcn_algo,
                                                  prepared=prepared)
    y0, s0 = oracle.lcn(x, 5, 0.05)
    if lcn_algo == "exact":
        assert np.array_equal(s.cpu().numpy(), s0), what + ": std is not the oracle's"
        assert np.array_equal(y.cpu().numpy(), y0), what + ": lcn is not the oracle's"
    else:
        assert_close(s.cpu().numpy(), s0, what=what + " fast std")
        assert_close(y.cpu().numpy(), y0, what=what + " fast lcn")
    # the matcher downstream of THIS lcn output: reference-order volume of y, its argmax
    vol_e = te.xcorrvol_batch(y, pat_lcn, D, 9, algo="exact")
    idx_e, best_e = te.argmax_disp(vol_e)
    bad = int((idx != idx_e).sum())
    assert bad == 0, "%s: %d of %d indices differ from the reference-order volume's" % (what, bad, idx_e.numel())
    assert_close(vol.cpu().numpy(), vol_e.cpu().numpy(), what=what + " volume")
    # the volume-free call returns the same indices
    _, _, idx_n, _ = te.lcn_xcorrvol_argmax(xd, pat_lcn, D, 9, 5, 0.05, lcn_algo=lcn_algo, prepared=prepared)
    assert torch.equal(idx_n, idx_e), what + ": volume-free indices"
    return y, idx


@pytest.mark.parametrize("N,H,W,D", [(1, 11, 16, 3), (2, 24, 232, 9), (1, 40, 236, 20), (3, 33, 300, 27), (1, 64, 464, 30),
                                     (2, 100, 512, 64), (1, 29, 1024, 40), (1, 13, 700, 130)])
@pytest.mark.parametrize("kind", ["uniform", "offset"])
def test_fused_exact_lcn_is_the_oracles_and_indices_are_the_references(te, oracle, N, H, W, D, kind):
    rs = np.random.RandomState(N * 7 + H + W + D)
    x = frames_of(kind, rs, N, H, W)
    pat = te.lcn(dev(workloads.syn_dot_pattern(H, W, seed=3)[None, None]), 5, 0.05)[0][0].contiguous()
    check(te, oracle, x, pat, D, "exact", what="%s %s" % ((N, H, W, D), kind))


@pytest.mark.parametrize("kind", ["dots", "dots_noise"])
def test_fused_exact_on_structured_light_frames(te, oracle, kind):
    """flat or low-noise background with sparse bright samples (the input the f32 LCN's centring is fragile on)"""
    rs = np.random.RandomState(5)
    x = frames_of(kind, rs, 2, 70, 300)
    pat = te.lcn(dev(workloads.syn_dot_pattern(70, 300, seed=3)[None, None]), 5, 0.05)[0][0].contiguous()
    check(te, oracle, x, pat, 40, "exact", what=kind)


@pytest.mark.parametrize("N,H,W,D", [(2, 24, 232, 9), (3, 33, 300, 27), (2, 100, 512, 64)])
def test_fused_fast_lcn_within_tolerance(te, oracle, N, H, W, D):
    rs = np.random.RandomState(N * 7 + H + W + D)
    pat = te.lcn(dev(workloads.syn_dot_pattern(H, W, seed=3)[None, None]), 5, 0.05)[0][0].contiguous()
    for kind in ("uniform", "dots"):
        check(te, oracle, frames_of(kind, rs, N, H, W), pat, D, "fast", what="%s %s fast" % ((N, H, W, D), kind))


def test_fused_prepared_pattern_and_config2_shape(te, oracle):
    """the bench step's call: 4 frames of 432 x 512, 128 disparities, prepared pattern, twice on one handle"""
    N, H, W, D = 4, 432, 512, 128
    x = np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])
    pat = te.lcn(dev(workloads.syn_dot_pattern(H, W, seed=42)[None, None]), 5, 0.05)[0][0].contiguous()
    prep = te.prepare_pattern(pat, N, D, 9)
    y1, idx1 = check(te, oracle, x, pat, D, "exact", prepared=prep, what="config 2 shape")
    y2, _, idx2, _ = te.lcn_xcorrvol_argmax(dev(x), pat, D, 9, prepared=prep)
    assert torch.equal(y1, y2) and torch.equal(idx1, idx2)
    # and the unfused pair of calls gives the same indices (same LCN bits -> same reference-order volume)
    ya, _ = te.lcn(dev(x), 5, 0.05)
    idx_u, _ = te.xcorrvol_argmax(ya, pat, D, 9)
    assert torch.equal(ya, y1) and torch.equal(idx_u, idx1)


def test_fused_falls_back_on_unsupported_shapes(te, oracle):
    """W % 4 != 0 or another radius: the two calls the fused one replaces run instead, same results"""
    rs = np.random.RandomState(2)
    x = rs.rand(1, 1, 30, 70).astype(np.float32)
    pat = te.lcn(dev(rs.rand(1, 1, 30, 70).astype(np.float32)), 5, 0.05)[0][0].contiguous()
    y, s, idx, best = te.lcn_xcorrvol_argmax(dev(x), pat, 12, 9)
    y0, s0 = oracle.lcn(x, 5, 0.05)
    assert np.array_equal(y.cpu().numpy(), y0) and np.array_equal(s.cpu().numpy(), s0)
    idx_e, _ = te.argmax_disp(te.xcorrvol_batch(y, pat, 12, 9, algo="exact"))
    assert torch.equal(idx, idx_e)


def test_fast_lcn_moves_near_tie_pixels_only_at_config2(te):
    """End-to-end index parity of the bench's timed step (16 x 432 x 512, 128 disparities, fused LCN with f32 box sums)
    against the pipeline raw frame -> lcn(algo='exact') (the oracle's bits) -> reference-order volume -> argmax: the LCN
    contract is a tolerance (networks.py:523-533: ATen's conv2d summation order is unspecified), so a pixel may change
    its index -- but only where the exact volume's best and the chosen score are closer than the LCN's tolerance can move
    an NCC score, and only a handful of the 3.5 M pixels do."""
    N, H, W, D = 16, 432, 512, 128
    x = dev(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)]))
    pat = te.lcn(dev(workloads.syn_dot_pattern(H, W, seed=42)[None, None]), 5, 0.05)[0][0].contiguous()
    prep = te.prepare_pattern(pat, N, D, 9)
    _, _, idx, _ = te.lcn_xcorrvol_argmax(x, pat, D, 9, 5, 0.05, lcn_algo="fast", prepared=prep)
    y_e, _ = te.lcn(x, 5, 0.05, algo="exact")
    vol_e = te.xcorrvol_batch(y_e, pat, D, 9, algo="exact")
    idx_e, best_e = te.argmax_disp(vol_e)
    moved = idx != idx_e
    n_moved = int(moved.sum())
    assert n_moved <= 64, "%d of %d pixels changed their index" % (n_moved, idx.numel())
    if n_moved:
        chosen = vol_e.gather(1, idx[:, None])[:, 0]           # the exact volume's score at the fused step's index
        gap = (best_e - chosen)[moved]
        # an LCN output within 1e-5 |y| + 1e-6 moves a zero-mean NCC score (81 taps, |score| <= 1) by a few 1e-5 at most
        assert float(gap.max()) <= 5e-5, "a pixel moved across a gap of %.3e" % float(gap.max())
    # with the oracle's LCN bits nothing moves
    _, _, idx_x, _ = te.lcn_xcorrvol_argmax(x, pat, D, 9, 5, 0.05, lcn_algo="exact", prepared=prep)
    assert torch.equal(idx_x, idx_e)
