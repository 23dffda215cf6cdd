"""BASELINE config 4: 1024 x 1024 frames, 256 disparities, block 9 -- the NCC volume (A1, exact and fast), its
argmax (A5) and the soft-census cost volume (A6, type census_sad, eps 0.5).  The serial oracle needs ~90 s per
full volume at this size, so parity against it is taken on horizontal strips: rows whose 9-row window does not
touch the strip border are computed from exactly the same taps as in the full image (row clamping only acts at
the image border, ext.h:148), so the oracle on a 12-row strip pins the 4 interior rows bit for bit.  The whole
volume is covered by size-independent properties."""
import numpy as np
import pytest
import torch

from tests import workloads
from tests.util import assert_close

pytestmark = pytest.mark.gpu

H = W = 1024
D = 256
BS = 9


@pytest.fixture(scope="module")
def te():
    from connecting_the_dots_amd import torchext
    return torchext


@pytest.fixture(scope="module")
def data(te):
    frame = workloads.uniform_frame(77, H, W)                       # [1,H,W]
    pat = workloads.syn_dot_pattern(H, W, seed=42)[None]            # [1,H,W] seeded dot pattern (commons.py:8-11)
    x, _ = te.lcn(torch.from_numpy(frame[None]).cuda(), 5, 0.05)
    p, _ = te.lcn(torch.from_numpy(pat[None]).cuda(), 5, 0.05)
    return x[0].contiguous(), p[0].contiguous()                     # [1,H,W] each, LCN'd as in exp_synph.py:64-91


STRIPS = [0, 500, H - 12]                                           # top border, interior, bottom border


def strip_rows(r0):
    """rows of the strip [r0, r0+12) whose window sees the same taps as in the full image"""
    lo = 0 if r0 == 0 else 4
    hi = 12 if r0 + 12 == H else 8
    return lo, hi


def test_exact_volume_strips_bit_exact(te, oracle, data):
    x, p = data
    vol = te.xcorrvol_batch(x[None], p, D, BS, algo="exact")[0]
    assert vol.shape == (D, H, W)
    xn, pn = x.cpu().numpy(), p.cpu().numpy()
    for r0 in STRIPS:
        ref = oracle.xcorrvol(xn[:, r0:r0 + 12], pn[:, r0:r0 + 12], D, BS, nthreads=8)
        lo, hi = strip_rows(r0)
        got = vol[:, r0 + lo:r0 + hi].cpu().numpy()
        assert np.array_equal(got, ref[:, lo:hi]), "strip at row %d" % r0


def test_fast_volume_tolerance_and_argmax_bit_parity(te, data):
    """fast (ranked) path against the reference-order HIP kernel at full size -- the checker is HIP too, but it is
    the kernel the previous test pins to the oracle bit for bit on strips of this very input (and to the reference's
    goldens in test_xcorrvol_gpu.py).  Also the volume-free ranked argmax at this size."""
    x, p = data
    exact = te.xcorrvol_batch(x[None], p, D, BS, algo="exact")[0]
    idx_f, best_f, fast = te.xcorrvol_argmax(x[None], p, D, BS, return_volume=True, algo="fast")
    # |a-b| <= 1e-5|b| + 1e-6 over all 268 M outputs
    err = (fast[0] - exact).abs()
    bound = exact.abs() * 1e-5 + 1e-6
    assert bool((err <= bound).all()), "max err %.3g" % err.max().item()
    del err, bound
    # indices: fast + exact re-rank == fused exact kernel == torch.argmax of the exact volume
    idx_e, best_e = te.xcorrvol_argmax(x[None], p, D, BS, algo="exact")
    assert torch.equal(idx_f, idx_e)
    assert torch.equal(idx_e[0], exact.argmax(0))
    assert torch.equal(best_e[0], exact.max(0).values)
    assert idx_f.dtype == torch.int64 and int(idx_f.max()) < D and int(idx_f.min()) >= 0
    del fast, exact
    idx_n, _ = te.xcorrvol_argmax(x[None], p, D, BS, algo="fast")             # nothing materialised
    assert torch.equal(idx_n, idx_e)


def test_self_match_is_one_at_zero_disparity(te, data):
    """xcorrvol(x, x) at d = 0 is the self-correlation: 1 up to rounding wherever the window is not flat"""
    x, _ = data
    idx, best = te.xcorrvol_argmax(x[None], x, D, BS, algo="fast")
    assert float((idx == 0).float().mean()) > 0.999
    b0 = best[idx == 0].cpu().numpy()
    assert_close(b0, np.ones_like(b0), rtol=1e-5, atol=1e-6, what="self NCC")


def test_census_cost_volume_strips_and_shift_recovery(te, oracle, data):
    x, p = data
    # frame = pattern shifted by a known disparity: the census cost volume's argmin must recover it
    true_d = 37
    cols = (torch.arange(W, device="cuda") - true_d).clamp(0, W - 1)
    shifted = p[0][:, cols].contiguous()
    cost = te.costvol(shifted, p[0], D, BS, "census_sad", 0.5, algo="exact")
    assert cost.shape == (D, H, W)
    arg = cost.argmin(0)
    inner = arg[8:-8, true_d + 8:-8]
    assert float((inner == true_d).float().mean()) > 0.999
    assert float(cost[true_d, 8:-8, true_d + 8:-8].abs().max()) == 0.0
    # strips against the oracle composition (photometric_loss_forward over shifted patterns), bit-exact
    sn, pn = shifted.cpu().numpy(), p[0].cpu().numpy()
    for r0 in STRIPS[:2]:
        ref = oracle.costvol(sn[r0:r0 + 12], pn[r0:r0 + 12], 48, BS, 3, 0.5, nthreads=8)
        lo, hi = strip_rows(r0)
        got = cost[:48, r0 + lo:r0 + hi].cpu().numpy()
        assert np.array_equal(got, ref[:, lo:hi]), "strip at row %d" % r0


@pytest.mark.parametrize("kind", ["census_sad", "sad"])
def test_lds_tiled_cost_volume_at_full_size(te, oracle, data, kind):
    """the LDS-tiled cost-volume kernel (algo='fast', the package default) at the config-4 size -- 8 disparity chunks
    of 32 per tile, frames * D = 256 grid planes: every one of the 268 M outputs within the float tolerance of the
    reference-order kernel (itself bit-identical to the oracle, test above), plus strips against the oracle directly"""
    x, p = data
    true_d = 37
    cols = (torch.arange(W, device="cuda") - true_d).clamp(0, W - 1)
    shifted = p[0][:, cols].contiguous()
    fast = te.costvol(shifted, p[0], D, BS, kind, 0.5, algo="fast")
    exact = te.costvol(shifted, p[0], D, BS, kind, 0.5, algo="exact")
    bad = (fast - exact).abs() > exact.abs() * 1e-5 + 1e-6
    assert int(bad.sum()) == 0, "%d outputs outside tolerance, max |a-b| %.3e" % (int(bad.sum()), float((fast - exact).abs().max()))
    del exact, bad
    assert torch.equal(fast.argmin(0)[8:-8, true_d + 8:-8], torch.full((H - 16, W - true_d - 16), true_d, device="cuda"))
    sn, pn = shifted.cpu().numpy(), p[0].cpu().numpy()
    ty = {"sad": 1, "census_sad": 3}[kind]
    for r0 in STRIPS:
        ref = oracle.costvol(sn[r0:r0 + 12], pn[r0:r0 + 12], 48, BS, ty, 0.5, nthreads=8)
        lo, hi = strip_rows(r0)
        assert_close(fast[:48, r0 + lo:r0 + hi].cpu().numpy(), ref[:, lo:hi], what="%s strip at row %d" % (kind, r0))


@pytest.mark.parametrize("kind", ["sad", "census_sad", "mse"])
def test_cost_volumes_on_the_bench_frame_at_full_size(te, oracle, data, kind):
    """the frame `bench.py --workload config4` times -- an LCN'd uniform random frame against the LCN'd dot pattern, not a
    shifted pattern: every one of the 268 M outputs of the fast cost volumes (separable SAD / MSE, census transform)
    within the float tolerance of the reference-order kernel, and strips against the oracle's composition directly"""
    x, p = data
    fast = te.costvol(x[0], p[0], D, BS, kind, 0.5, algo="fast")
    exact = te.costvol(x[0], p[0], D, BS, kind, 0.5, algo="exact")
    bad = (fast - exact).abs() > exact.abs() * 1e-5 + 1e-6
    assert int(bad.sum()) == 0, "%d outputs outside tolerance, max |a-b| %.3e" % (int(bad.sum()), float((fast - exact).abs().max()))
    del exact, bad
    xn, pn = x[0].cpu().numpy(), p[0].cpu().numpy()
    ty = {"mse": 0, "sad": 1, "census_sad": 3}[kind]
    for r0 in STRIPS[:2]:
        ref = oracle.costvol(xn[r0:r0 + 12], pn[r0:r0 + 12], 48, BS, ty, 0.5, nthreads=8)
        lo, hi = strip_rows(r0)
        assert_close(fast[:48, r0 + lo:r0 + hi].cpu().numpy(), ref[:, lo:hi], what="%s strip at row %d" % (kind, r0))
