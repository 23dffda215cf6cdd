"""GPU parity of the separable-sum NCC volume (algo='fast'): float tolerance of SURVEY 8d,
|a-b| <= 1e-5*|b| + 1e-6, against the reference-order oracle / goldens."""
import numpy as np
import pytest
import torch

from tests import workloads
from tests.util import assert_close, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def te():
    from connecting_the_dots_amd import torchext
    return torchext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_fast_small_goldens(te):
    g = golden("xcorrvol_small")
    n = 0
    for k, (C, H, W, D, bs) in enumerate(g["cases"]):
        if g["in0_%d" % k].dtype != np.float32 or bs % 2 == 0:
            continue
        n += 1
        vol = te.xcorrvol_batch(dev(g["in0_%d" % k][None]), dev(g["in1_%d" % k]), int(D), int(bs), algo="fast")
        assert_close(vol[0].cpu().numpy(), g["vol_%d" % k], what="case %d" % k)
    assert n >= 4


@pytest.mark.parametrize("shape", [(1, 7, 9, 5, 9), (2, 33, 130, 37, 9), (1, 5, 64, 128, 9), (1, 40, 200, 19, 7),
                                   (1, 11, 70, 9, 5), (1, 11, 70, 9, 3), (1, 150, 57, 16, 9), (1, 8, 300, 256, 9)])
@pytest.mark.parametrize("kind", ["normal", "uniform", "offset"])
def test_fast_vs_oracle_ragged(te, oracle, shape, kind):
    C, H, W, D, bs = shape
    rs = np.random.RandomState(sum(shape))
    if kind == "normal":
        a, b = rs.randn(C, H, W), rs.randn(C, H, W)
    elif kind == "uniform":
        a, b = rs.rand(C, H, W), rs.rand(C, H, W)
    else:                                   # 8-bit style intensities with a large DC offset
        a, b = rs.rand(C, H, W) * 60 + 150, rs.rand(C, H, W) * 90 + 100
    a, b = a.astype(np.float32), b.astype(np.float32)
    ref = oracle.xcorrvol(a, b, D, bs, nthreads=8)
    vol = te.xcorrvol_batch(dev(a[None]), dev(b), D, bs, algo="fast")[0].cpu().numpy()
    assert_close(vol, ref, what="%s %s" % (shape, kind))


def test_fast_left_border_ties_stay_exact(te):
    """Once w - d + 4 < 0 every pattern tap clamps to column 0 (ext.h:152-154): the reference volume is
    exactly constant in d there; the fast kernel must keep those ties exact (first index wins later)."""
    rs = np.random.RandomState(3)
    a = rs.randn(1, 1, 20, 64).astype(np.float32)
    b = rs.randn(1, 20, 64).astype(np.float32)
    vol = te.xcorrvol_batch(dev(a), dev(b), 32, 9, algo="fast")[0].cpu().numpy()
    for w in range(0, 20):
        d0 = w + 4                                  # all d >= d0 see the same clamped window
        if d0 + 1 < 32:
            assert (vol[d0:, :, w] == vol[d0, :, w]).all(), w


def test_fast_cfg1_full_size(te):
    g = golden("xcorrvol_cfg1")
    a = workloads.uniform_frame(1234, 432, 512)
    b = workloads.uniform_frame(42, 432, 512)
    vol = te.xcorrvol_batch(dev(a[None]), dev(b), 128, 9, algo="fast")[0]
    got = vol.reshape(-1)[torch.from_numpy(g["sample_idx"]).cuda()].cpu().numpy()
    assert_close(got, g["uni_sample_val"], what="cfg1 uniform samples")
    # algorithm-independent property: NCC of a frame with itself is 1 at d = 0
    self_vol = te.xcorrvol_batch(dev(a[None]), dev(a), 4, 9, algo="fast")[0, 0].cpu().numpy()
    assert np.abs(self_vol - 1).max() < 1e-5


def test_fast_multichannel_and_per_frame_pattern(te, oracle):
    rs = np.random.RandomState(12)
    a = rs.randn(2, 3, 24, 70).astype(np.float32)
    b = rs.randn(2, 3, 24, 70).astype(np.float32)
    vol = te.xcorrvol_batch(dev(a), dev(b), 20, 9, algo="fast").cpu().numpy()
    for f in range(2):
        assert_close(vol[f], oracle.xcorrvol(a[f], b[f], 20, 9, nthreads=4), rtol=1e-5, atol=3e-6, what="frame %d" % f)


def test_fast_argmax_rerank_matches_reference_indices(te, oracle):
    """algo='fast' + re-rank: indices bit-identical to torch.argmax of the reference-order volume."""
    g = golden("xcorrvol_small")
    n = 0
    for k, (C, H, W, D, bs) in enumerate(g["cases"]):
        if C != 1 or g["in0_%d" % k].dtype != np.float32 or bs % 2 == 0:
            continue
        n += 1
        idx, best = te.xcorrvol_argmax(dev(g["in0_%d" % k]), dev(g["in1_%d" % k]), int(D), int(bs), algo="fast")
        assert np.array_equal(idx.cpu().numpy(), g["argmax_%d" % k]), "case %d" % k
        assert_close(best.cpu().numpy(), g["vol_%d" % k].max(0), what="best %d" % k)
    assert n >= 3


def test_fast_argmax_cfg1_full_size_bit_parity(te):
    """BASELINE config 1/2 shape: disparity MAE vs reference == 0 on both golden workloads."""
    g = golden("xcorrvol_cfg1")
    a = workloads.uniform_frame(1234, 432, 512)
    b = workloads.uniform_frame(42, 432, 512)
    idx, best = te.xcorrvol_argmax(dev(a), dev(b), 128, 9, algo="fast")
    assert np.array_equal(idx.cpu().numpy().astype(np.uint8), g["uni_argmax"])


def test_fast_argmax_kinect_bit_parity(te, oracle):
    g = golden("xcorrvol_cfg1")
    pat = g["kin_pattern_u8"].astype(np.float32) / 255
    ir, _ = workloads.synth_ir(pat, np.random.RandomState(2024), 128)
    ir_l, _ = oracle.lcn(ir[None, None], 5, 0.05)
    pat_l, _ = oracle.lcn(pat[None, None], 5, 0.05)
    idx, best = te.xcorrvol_argmax(dev(ir_l[0]), dev(pat_l[0]), 128, 9, algo="fast")
    assert np.array_equal(idx.cpu().numpy().astype(np.uint8), g["kin_argmax"])


@pytest.mark.parametrize("C,per_frame", [(1, False), (2, False), (1, True)])
def test_fast_ill_conditioned_windows_are_recomputed(te, oracle, C, per_frame):
    """Flat regions, windows clamped onto a flat border column and smooth backgrounds with a DC offset make
    cov = S_ab - n*ma*mb cancel in f32; the fast path lists such windows and recomputes their outputs in
    reference order (ncc_fixup_kernel), so the tolerance holds for any input, not only LCN'd frames."""
    rs = np.random.RandomState(17 + C)
    N, H, W, D, bs = 2, 40, 96, 40, 9
    yy, xx = np.mgrid[0:H, 0:W]
    bg = (100 + 60 * np.sin(xx / 23.0) * np.cos(yy / 17.0)).astype(np.float32)
    a = (rs.rand(N, C, H, W) * 8 + bg).astype(np.float32)
    a[:, :, 5:22, 30:60] = 0.25                      # exactly flat block: sigma0 == 0 windows
    b = (rs.rand(N if per_frame else 1, C, H, W) * 20 + 0.6 * bg).astype(np.float32)
    b[:, :, 10:30, 0:3] = 7.0                        # flat left border: the fully clamped run x <= -4 is degenerate
    b[:, :, 24:38, 50:70] = 0.0                      # flat interior block
    bb = dev(b) if per_frame else dev(b[0])
    vol = te.xcorrvol_batch(dev(a), bb, D, bs, algo="fast").cpu().numpy()
    for f in range(N):
        ref = oracle.xcorrvol(a[f], b[f if per_frame else 0], D, bs, nthreads=8)
        assert_close(vol[f], ref, what="frame %d" % f)
    if C == 1:
        idx, best = te.xcorrvol_argmax(dev(a), bb, D, bs, algo="fast")
        idx_e, best_e = te.xcorrvol_argmax(dev(a), bb, D, bs, algo="exact")
        assert torch.equal(idx, idx_e)


def test_fast_constant_images(te, oracle):
    """every window flagged: the whole volume comes out of the reference-order fix-up"""
    a = np.full((1, 1, 24, 64), 0.5, np.float32)
    b = np.full((1, 24, 64), 0.25, np.float32)
    b[0, 10, 20] = 1.0
    vol = te.xcorrvol_batch(dev(a), dev(b), 16, 9, algo="fast")[0].cpu().numpy()
    assert np.array_equal(vol, oracle.xcorrvol(a[0], b, 16, 9))


def test_fast_random_shapes_vs_exact(te):
    """seeded sweep over shapes / channel counts / block sizes / input kinds (flat halves, DC offsets, dot patterns,
    per-frame patterns; W not a multiple of 4 takes the fallback kernels): volume within tolerance of the
    reference-order kernel and, for C == 1, re-ranked indices identical to the fused exact argmax
    (tools/fuzz_fast.py runs the same check over hundreds of configurations).  The checker here is the
    reference-order HIP kernel, not the oracle: it is pinned to the reference bit for bit in test_xcorrvol_gpu.py
    (small goldens incl. f64 / even blocks, two full-size SHA-256s) and in test_config4_gpu.py (oracle strips)."""
    rs = np.random.RandomState(11)
    for it in range(24):
        C = int(rs.choice([1, 1, 1, 2, 3])); N = int(rs.randint(1, 4)); bs = int(rs.choice([9, 9, 9, 7, 5, 3]))
        H = int(rs.randint(1, 90)); W = int(rs.choice([rs.randint(1, 70), 4 * rs.randint(1, 100), rs.randint(60, 400)]))
        D = int(rs.choice([1, 2, rs.randint(1, 40), rs.randint(40, 200)]))
        per_frame = bool(rs.rand() < 0.3)
        kind = rs.choice(["normal", "flat", "offset", "dots"])
        a = rs.randn(N, C, H, W).astype(np.float32)
        b = rs.randn(N if per_frame else 1, C, H, W).astype(np.float32)
        if kind == "flat":
            a[:, :, : H // 2] = 0.5
            b[:, :, :, : max(1, W // 3)] = -1.0
        if kind == "offset":
            a, b = a * 5 + 100, b * 9 + 40
        if kind == "dots":
            b = (rs.rand(*b.shape) < 0.1).astype(np.float32)
        A, B = dev(a), dev(b if per_frame else b[0])
        ex = te.xcorrvol_batch(A, B, D, bs, algo="exact")
        fa = te.xcorrvol_batch(A, B, D, bs, algo="fast")
        cfg = (N, C, H, W, D, bs, per_frame, str(kind))
        assert bool(((fa - ex).abs() <= ex.abs() * 1e-5 + 1e-6).all()), cfg
        if C == 1:
            assert torch.equal(te.xcorrvol_argmax(A, B, D, bs, algo="fast")[0], te.xcorrvol_argmax(A, B, D, bs, algo="exact")[0]), cfg
