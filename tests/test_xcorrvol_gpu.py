"""GPU parity of the NCC volume / argmax against the CPU oracle and the committed goldens.
Everything goes through the C ABI (ctypes) via connecting_the_dots_amd.torchext."""
import hashlib

import numpy as np
import pytest
import torch

from tests import workloads
from tests.util import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def reference_order_kernels(monkeypatch):
    """this module pins the reference-order kernels (bit-exact asserts): the package default is algo='fast'"""
    monkeypatch.setenv("CTD_NCC_ALGO", "exact")
    monkeypatch.setenv("CTD_PHOTO_ALGO", "exact")


@pytest.fixture(scope="module")
def te():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from connecting_the_dots_amd import torchext
    return torchext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_xcorrvol_small_golden_bit_exact(te):
    g = golden("xcorrvol_small")
    for k, (C, H, W, D, bs) in enumerate(g["cases"]):
        vol = te.xcorrvol(dev(g["in0_%d" % k]), dev(g["in1_%d" % k]), int(D), int(bs)).cpu().numpy()
        assert vol.dtype == g["vol_%d" % k].dtype
        assert np.array_equal(vol, g["vol_%d" % k]), "case %d (C=%d bs=%d): max diff %g" % (
            k, C, bs, np.abs(vol - g["vol_%d" % k]).max())


def test_argmax_first_index_wins_ties(te):
    g = golden("xcorrvol_small")
    for k in range(len(g["cases"])):
        if g["vol_%d" % k].dtype != np.float32:
            continue
        idx, best = te.argmax_disp(dev(g["vol_%d" % k]))
        assert np.array_equal(idx.cpu().numpy(), g["argmax_%d" % k])
        assert np.array_equal(best.cpu().numpy(), g["vol_%d" % k].max(0))


def test_fused_argmax_small(te):
    g = golden("xcorrvol_small")
    n = 0
    for k, (C, H, W, D, bs) in enumerate(g["cases"]):
        if C != 1 or g["in0_%d" % k].dtype != np.float32 or bs % 2 == 0:
            continue
        n += 1
        idx, best, vol = te.xcorrvol_argmax(dev(g["in0_%d" % k]), dev(g["in1_%d" % k]), int(D), int(bs),
                                            return_volume=True, algo="exact")
        assert np.array_equal(vol.cpu().numpy(), g["vol_%d" % k])
        assert np.array_equal(idx.cpu().numpy(), g["argmax_%d" % k])
        idx2, best2 = te.xcorrvol_argmax(dev(g["in0_%d" % k]), dev(g["in1_%d" % k]), int(D), int(bs), algo="exact")
        assert np.array_equal(idx2.cpu().numpy(), g["argmax_%d" % k])
        assert np.array_equal(best2.cpu().numpy(), g["vol_%d" % k].max(0))
    assert n >= 3


@pytest.mark.parametrize("shape", [(1, 7, 9, 5, 9), (2, 33, 130, 37, 9), (1, 5, 64, 128, 9), (1, 40, 200, 19, 7),
                                   (1, 11, 70, 9, 5), (1, 11, 70, 9, 3), (1, 13, 21, 6, 6), (1, 8, 300, 256, 9)])
def test_xcorrvol_vs_oracle_ragged(te, oracle, shape):
    C, H, W, D, bs = shape
    rs = np.random.RandomState(sum(shape))
    a = rs.randn(C, H, W).astype(np.float32)
    b = rs.randn(C, H, W).astype(np.float32)
    ref = oracle.xcorrvol(a, b, D, bs, nthreads=8)
    vol = te.xcorrvol(dev(a), dev(b), D, bs).cpu().numpy()
    assert np.array_equal(vol, ref), "max diff %g" % np.abs(vol - ref).max()


def test_xcorrvol_f64_vs_oracle(te, oracle):
    rs = np.random.RandomState(5)
    a, b = rs.rand(2, 12, 40), rs.rand(2, 12, 40)
    ref = oracle.xcorrvol(a, b, 10, 9, nthreads=4)
    vol = te.xcorrvol(dev(a), dev(b), 10, 9).cpu().numpy()
    assert np.array_equal(vol, ref)


def test_xcorrvol_generic_kernel_chunks_the_disparities(te, oracle):
    """f64, block 15, D = 600: the pattern rows of all disparities (18 x 755 doubles = 109 KB beside the frame tile) do
    not fit the generic kernel's 64 KB of LDS -- the disparities are served in chunks (round 2 returned
    CTD_ERR_UNSUPPORTED from 160 KB on); two channels, so that the per-chunk channel accumulation is exercised too"""
    rs = np.random.RandomState(15)
    a, b = rs.rand(2, 9, 70), rs.rand(2, 9, 70)
    ref = oracle.xcorrvol(a, b, 600, 15, nthreads=8)
    vol = te.xcorrvol(dev(a), dev(b), 600, 15).cpu().numpy()
    assert np.array_equal(vol, ref)
    a32, b32 = a.astype(np.float32), b.astype(np.float32)           # f32, even block: the same kernel
    ref32 = oracle.xcorrvol(a32, b32, 1100, 12, nthreads=8)
    vol32 = te.xcorrvol(dev(a32), dev(b32), 1100, 12, algo="exact").cpu().numpy()
    assert np.array_equal(vol32, ref32)


def test_xcorrvol_batch_shared_and_per_frame_pattern(te, oracle):
    rs = np.random.RandomState(8)
    a = rs.randn(3, 1, 20, 96).astype(np.float32)
    b = rs.randn(3, 1, 20, 96).astype(np.float32)
    shared = te.xcorrvol_batch(dev(a), dev(b[0]), 24, 9).cpu().numpy()
    per = te.xcorrvol_batch(dev(a), dev(b), 24, 9).cpu().numpy()
    for f in range(3):
        assert np.array_equal(shared[f], oracle.xcorrvol(a[f], b[0], 24, 9, nthreads=4))
        assert np.array_equal(per[f], oracle.xcorrvol(a[f], b[f], 24, 9, nthreads=4))


def test_cfg1_uniform_full_size_sha256(te):
    """BASELINE config 1 at full size: 512x432x128 volume, SHA-256 equal to the reference's."""
    g = golden("xcorrvol_cfg1")
    a = workloads.uniform_frame(1234, 432, 512)
    b = workloads.uniform_frame(42, 432, 512)
    idx, best, vol = te.xcorrvol_argmax(dev(a), dev(b), 128, 9, return_volume=True, algo="exact")
    vol = vol.cpu().numpy()
    assert np.array_equal(vol.reshape(-1)[g["sample_idx"]], g["uni_sample_val"])
    assert hashlib.sha256(vol.tobytes()).digest() == g["uni_sha256"].tobytes()
    assert np.array_equal(idx.cpu().numpy().astype(np.uint8), g["uni_argmax"])       # disparity MAE vs ref == 0
    vol2 = te.xcorrvol(dev(a), dev(b), 128, 9)
    assert torch.equal(vol2.cpu(), torch.from_numpy(vol))


def test_cfg1_kinect_pattern_sha256(te, oracle):
    g = golden("xcorrvol_cfg1")
    pat = g["kin_pattern_u8"].astype(np.float32) / 255
    ir, _ = workloads.synth_ir(pat, np.random.RandomState(2024), 128)
    ir_l, _ = oracle.lcn(ir[None, None], 5, 0.05)
    pat_l, _ = oracle.lcn(pat[None, None], 5, 0.05)
    idx, best, vol = te.xcorrvol_argmax(dev(ir_l[0]), dev(pat_l[0]), 128, 9, return_volume=True, algo="exact")
    assert hashlib.sha256(vol.cpu().numpy().tobytes()).digest() == g["kin_sha256"].tobytes()
    assert np.array_equal(idx.cpu().numpy().astype(np.uint8), g["kin_argmax"])


def test_error_behaviour(te):
    a = torch.rand(1, 8, 8)
    with pytest.raises(RuntimeError):
        te.xcorrvol(a, a, 4, 9)                                  # CPU tensor: no CPU path here
    c = torch.rand(1, 8, 16).cuda()[:, :, ::2]
    with pytest.raises(RuntimeError):
        te.xcorrvol(c, c, 4, 9)                                  # CHECK_CONTIGUOUS (ext.h:7)
    x = torch.rand(1, 8, 8).cuda().requires_grad_(True)
    out = te.xcorrvol(x, torch.rand(1, 8, 8).cuda(), 4, 9)
    out.sum().backward() if out.requires_grad else None          # backward returns None (functions.py:70-71)
    assert x.grad is None
