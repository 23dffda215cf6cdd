"""GPU parity of the in-kernel ranking of the fast NCC path (ncc_fast.hip all-D kernel -> fix-up -> tail kernel):
indices must equal torch.argmax of the reference-order volume bit for bit, with the volume
materialised (return_volume=True) and without one (volume-free).  The checker is the reference-order HIP kernel
(`algo='exact'`), itself pinned bit for bit to the reference's goldens in test_xcorrvol_gpu.py, and the CPU oracle
for the small cases."""
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


def check_both_modes(te, A, B, D, what, exact=None):
    """fast + rank (with and without a volume) vs the reference-order kernel"""
    if exact is None:
        exact = te.xcorrvol_argmax(A, B, D, 9, return_volume=True, algo="exact")
    idx_e, best_e, vol_e = exact
    idx_v, best_v, vol_v = te.xcorrvol_argmax(A, B, D, 9, return_volume=True, algo="fast")
    idx_n, best_n = te.xcorrvol_argmax(A, B, D, 9, algo="fast")
    bad_v, bad_n = int((idx_v != idx_e).sum()), int((idx_n != idx_e).sum())
    assert bad_v == 0 and bad_n == 0, "%s: %d / %d of %d indices differ (volume / volume-free)" % (
        what, bad_v, bad_n, idx_e.numel())
    # the materialised volume is the plain fast volume (ranking must not disturb it)
    plain = te.xcorrvol_batch(A if A.dim() == 4 else A[None], B, D, 9, algo="fast")
    assert torch.equal(vol_v if vol_v.dim() == 4 else vol_v[None], plain), what
    tol = vol_e.abs().amax(-3) * 1e-5 + 2e-6          # fast score + key resolution (2^-21 absolute)
    assert bool(((best_v - best_e).abs() <= tol).all()), what
    assert bool(((best_n - best_e).abs() <= tol).all()), what
    return idx_e


@pytest.mark.parametrize("N,H,W,D", [(1, 9, 4, 1), (1, 3, 8, 2), (2, 20, 64, 13), (1, 31, 256, 14), (1, 17, 260, 15),
                                     (3, 45, 300, 27), (1, 50, 512, 28), (1, 12, 516, 129), (2, 67, 128, 130),
                                     (1, 10, 1024, 256), (1, 6, 72, 300)])
def test_rank_shapes(te, N, H, W, D):
    rs = np.random.RandomState(N * 1000 + H + W + D)
    A = dev(rs.randn(N, 1, H, W).astype(np.float32))
    B = dev(rs.randn(1, H, W).astype(np.float32))
    check_both_modes(te, A, B, D, (N, H, W, D))


def test_rank_small_goldens_vs_reference(te):
    """indices of the reference itself (ext_cpu goldens), through both rank modes"""
    g = golden("xcorrvol_small")
    n = 0
    for k, (C, H, W, D, bs) in enumerate(g["cases"]):
        if C != 1 or g["in0_%d" % k].dtype != np.float32 or bs != 9 or W % 4:
            continue
        n += 1
        A, B = dev(g["in0_%d" % k]), dev(g["in1_%d" % k])
        for rv in (False, True):
            out = te.xcorrvol_argmax(A, B, int(D), 9, return_volume=rv, algo="fast")
            assert np.array_equal(out[0].cpu().numpy(), g["argmax_%d" % k]), (k, rv)
            assert_close(out[1].cpu().numpy(), g["vol_%d" % k].max(0), atol=2e-6, what="best %d" % k)
    assert n >= 1


def test_rank_per_frame_pattern_and_dots(te):
    rs = np.random.RandomState(5)
    N, H, W, D = 3, 40, 128, 64
    a = rs.randn(N, 1, H, W).astype(np.float32)
    b = (rs.rand(N, 1, H, W) < 0.1).astype(np.float32)       # raw dot patterns: many flat windows, exact ties
    check_both_modes(te, dev(a), dev(b), D, "per-frame dots")
    check_both_modes(te, dev(a), dev(b[0]), D, "shared dots")


def test_rank_listed_runs_and_flat_blocks(te, oracle):
    """flat left border (listed fully clamped run), flat interior blocks on both sides, DC offset: the ranking sees
    unpatched scores there; dirty pixels and the run candidate must bring the reference's indices back"""
    rs = np.random.RandomState(23)
    N, H, W, D = 2, 40, 96, 40
    yy, xx = np.mgrid[0:H, 0:W]
    bg = (100 + 60 * np.sin(xx / 23.0) * np.cos(yy / 17.0)).astype(np.float32)
    a = (rs.rand(N, 1, H, W) * 8 + bg).astype(np.float32)
    a[:, :, 5:22, 30:60] = 0.25
    b = (rs.rand(1, H, W) * 20 + 0.6 * bg).astype(np.float32)
    b[:, 10:30, 0:3] = 7.0
    b[:, 24:38, 50:70] = 0.0
    idx_e = check_both_modes(te, dev(a), dev(b), D, "listed")
    ref = np.stack([oracle.xcorrvol(a[f], b, D, 9, nthreads=8) for f in range(N)])
    assert np.array_equal(idx_e.cpu().numpy(), ref.argmax(1))


def test_rank_constant_and_periodic(te):
    a = np.full((1, 1, 24, 64), 0.5, np.float32)
    b = np.full((1, 24, 64), 0.25, np.float32)
    b[0, 10, 20] = 1.0
    check_both_modes(te, dev(a), dev(b), 16, "constant")
    # period-8 stripes: exact ties between disparities 8 apart everywhere (first index must win)
    xx = np.arange(128, dtype=np.float32)
    s = np.sin(2 * np.pi * xx / 8)[None, None, :].repeat(30, 1).astype(np.float32)
    n = np.random.RandomState(1).randn(1, 30, 128).astype(np.float32) * 1e-3
    check_both_modes(te, dev((s + n)[None]), dev(s + n), 40, "periodic")
    check_both_modes(te, dev(s[None]), dev(s), 40, "periodic exact")


def test_rank_synthetic_dot_pattern_lcn(te):
    """the bench workload at reduced height: LCN'd uniform frames against the LCN'd seeded dot pattern (flat
    column-0 windows -> listed runs in a third of the rows)"""
    H, W, D = 96, 512, 128
    fr = np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(2)])
    pat = workloads.syn_dot_pattern(H, W, seed=42)[None, None]
    x, _ = te.lcn(dev(fr), 5, 0.05)
    p, _ = te.lcn(dev(pat), 5, 0.05)
    check_both_modes(te, x, p[0].contiguous(), D, "dot pattern")


def test_rank_full_size_goldens(te, oracle):
    """BASELINE config 1/2 shape: disparity MAE vs the reference == 0 on both golden workloads, both modes"""
    g = golden("xcorrvol_cfg1")
    a = workloads.uniform_frame(1234, 432, 512)
    b = workloads.uniform_frame(42, 432, 512)
    for rv in (False, True):
        idx = te.xcorrvol_argmax(dev(a), dev(b), 128, 9, return_volume=rv, algo="fast")[0]
        assert np.array_equal(idx.cpu().numpy().astype(np.uint8), g["uni_argmax"]), rv
    pat = g["kin_pattern_u8"].astype(np.float32) / 255
    ir, _ = workloads.synth_ir(pat, np.random.RandomState(2024), 128)
    ir_l, _ = oracle.lcn(ir[None, None], 5, 0.05)
    pat_l, _ = oracle.lcn(pat[None, None], 5, 0.05)
    for rv in (False, True):
        idx = te.xcorrvol_argmax(dev(ir_l[0]), dev(pat_l[0]), 128, 9, return_volume=rv, algo="fast")[0]
        assert np.array_equal(idx.cpu().numpy().astype(np.uint8), g["kin_argmax"]), rv


def test_rank_config2_batch_one_round_plan(te):
    """BASELINE config 2 as the bench runs it (16 frames: ONE round of 256 workgroups, 54-row bands, two-row chunks -- a
    plan no smaller shape selects): volume + ranking, ranking alone and the plain volume against the reference-order
    kernel on the bench workload, every pixel of every frame"""
    from connecting_the_dots_amd import _lib
    import ctypes
    H, W, D, N = 432, 512, 128, 16
    off = (ctypes.c_size_t * 5)()
    assert _lib.lib().ctd_xcorrvol_rank_layout(N, H, W, D, 0, off) == 0 and (off[0], off[4]) == (54, 5)
    fr = np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])
    pat = workloads.syn_dot_pattern(H, W, seed=42)[None, None]
    x, _ = te.lcn(dev(fr), 5, 0.05)
    p = te.lcn(dev(pat), 5, 0.05)[0][0].contiguous()
    check_both_modes(te, x, p, D, "config 2, 16 frames")


def test_rank_eps_negative_is_plain_fast_argmax(te):
    """rerank_eps < 0: no re-scoring; the key-ranked index must then hold a score within the key truncation of the
    fast volume's maximum"""
    rs = np.random.RandomState(9)
    A, B = dev(rs.randn(2, 1, 30, 256).astype(np.float32)), dev(rs.randn(1, 30, 256).astype(np.float32))
    idx, best, vol = te.xcorrvol_argmax(A, B, 64, 9, return_volume=True, algo="fast", rerank_eps=-1.0)
    picked = vol.gather(1, idx[:, None])[:, 0]
    # scores of the clamped run are copies: compare values, not indices
    assert bool((vol.amax(1) - picked <= vol.amax(1).abs() * 4e-6 + 1e-7).all())


def test_rank_eps_negative_with_listed_windows(te):
    """rerank_eps < 0 on input with flat blocks and a listed fully clamped run (round-2 advice): the plain argmax must
    be that of the returned (patched) volume -- scores of listed windows exist only there"""
    rs = np.random.RandomState(31)
    N, H, W, D = 2, 40, 96, 40
    yy, xx = np.mgrid[0:H, 0:W]
    bg = (100 + 60 * np.sin(xx / 23.0) * np.cos(yy / 17.0)).astype(np.float32)
    a = (rs.rand(N, 1, H, W) * 8 + bg).astype(np.float32)
    a[:, :, 5:22, 30:60] = 0.25
    b = (rs.rand(1, H, W) * 20 + 0.6 * bg).astype(np.float32)
    b[:, 10:30, 0:3] = 7.0
    b[:, 24:38, 50:70] = 0.0
    idx, best, vol = te.xcorrvol_argmax(dev(a), dev(b), D, 9, return_volume=True, algo="fast", rerank_eps=-1.0)
    assert not bool(torch.isnan(vol).any())
    assert int(idx.min()) >= 0 and int(idx.max()) < D
    assert torch.equal(idx, vol.argmax(1)), int((idx != vol.argmax(1)).sum())
    assert torch.equal(best, vol.amax(1))
    # without a volume a negative eps means 0: exact re-scoring of what lies inside the key resolution -- reference indices
    idx_n, _ = te.xcorrvol_argmax(dev(a), dev(b), D, 9, algo="fast", rerank_eps=-1.0)
    idx_e = te.xcorrvol_argmax(dev(a), dev(b), D, 9, algo="exact")[0]
    assert int((idx_n != idx_e).sum()) <= int(0.001 * idx_e.numel())        # (eps 0: no guarantee, nearly all agree)


def test_prepared_pattern_gives_the_same_results(te):
    """prepare_pattern once, then several calls: index, best score and volume bit-identical to the unprepared call -- on
    the LCN'd dot pattern (listed clamped runs: the fix-up lists of the pattern must survive between calls), with and
    without a volume, shared and per-frame patterns"""
    H, W, D = 60, 512, 128
    fr = np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(3)])
    pat = workloads.syn_dot_pattern(H, W, seed=42)[None, None]
    x, _ = te.lcn(dev(fr), 5, 0.05)
    p = te.lcn(dev(pat), 5, 0.05)[0][0].contiguous()
    ref = te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast")
    h = te.prepare_pattern(p, 3, D, 9)
    for rep in range(3):
        x2 = (x + 0.0) if rep < 2 else x.flip(0).contiguous()        # the last call: other frames, same handle
        got = te.xcorrvol_argmax(x2, p, D, 9, return_volume=True, algo="fast", prepared=h)
        want = ref if rep < 2 else te.xcorrvol_argmax(x2, p, D, 9, return_volume=True, algo="fast")
        assert all(torch.equal(a, b) for a, b in zip(got, want)), rep
        got_n = te.xcorrvol_argmax(x2, p, D, 9, algo="fast", prepared=h)
        assert torch.equal(got_n[0], want[0])
        assert torch.equal(te.xcorrvol_batch(x2, p, D, 9, algo="fast", prepared=h), want[2])    # the plain volume call too
    pp = te.lcn(dev(np.stack([pat[0], pat[0][:, ::-1].copy(), pat[0]])), 5, 0.05)[0].contiguous()   # per-frame patterns
    h2 = te.prepare_pattern(pp, 3, D, 9)
    got = te.xcorrvol_argmax(x, pp, D, 9, return_volume=True, algo="fast", prepared=h2)
    want = te.xcorrvol_argmax(x, pp, D, 9, return_volume=True, algo="fast")
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    with pytest.raises(RuntimeError):
        te.xcorrvol_argmax(x[:2].contiguous(), p, D, 9, algo="fast", prepared=h)      # another frame count


@pytest.mark.parametrize("N,H,W,D", [(1, 21, 260, 130), (2, 40, 516, 33), (5, 12, 64, 256), (1, 9, 8, 3)])
def test_prepared_pattern_shapes(te, N, H, W, D):
    """prepared == unprepared on ragged shapes (D past one 128-disparity round, widths off the 256-column tiles, one
    frame), ranked with / without a volume and plain, twice in a row on the same handle"""
    rs = np.random.RandomState(N + H + W + D)
    A = dev(rs.randn(N, 1, H, W).astype(np.float32))
    flat = rs.randn(1, H, W).astype(np.float32)
    flat[:, H // 3:H // 3 + 9, :12] = 0.25                       # a flat patch at the left border: listed windows and runs
    B = dev(flat)
    want = te.xcorrvol_argmax(A, B, D, 9, return_volume=True, algo="fast")
    h = te.prepare_pattern(B, N, D, 9)
    for _ in range(2):
        got = te.xcorrvol_argmax(A, B, D, 9, return_volume=True, algo="fast", prepared=h)
        assert all(torch.equal(a, b) for a, b in zip(got, want))
        assert torch.equal(te.xcorrvol_argmax(A, B, D, 9, algo="fast", prepared=h)[0], want[0])
        assert torch.equal(te.xcorrvol_batch(A, B, D, 9, algo="fast", prepared=h), want[2])


@pytest.mark.parametrize("W,bs", [(261, 9), (258, 5), (67, 7)])
def test_workspace_contents_do_not_matter_fallback_kernels(te, W, bs):
    """the same for widths off the 4-column grid and the small block sizes (wide / narrow fallback kernels)"""
    from connecting_the_dots_amd import _lib
    L = _lib.lib()
    N, H, D = 2, 30, 37
    rs = np.random.RandomState(W + bs)
    A = dev(rs.randn(N, 1, H, W).astype(np.float32)); B = dev(rs.randn(1, H, W).astype(np.float32))
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for fill in (0x00, 0xFF):
        vol = torch.empty((N, D, H, W), device="cuda")
        ws = torch.full((L.ctd_xcorrvol_workspace_bytes(N, 1, H, W, D, bs, 1),), fill, dtype=torch.uint8, device="cuda")
        assert L.ctd_xcorrvol_f32(A.data_ptr(), B.data_ptr(), 0, vol.data_ptr(), N, 1, H, W, D, bs, 1, ws.data_ptr(), ws.numel(), 0, s) == 0
        outs.append(vol)
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[1]).all())
    ve = te.xcorrvol_batch(A, B, D, bs, algo="exact")
    assert bool(((outs[1] - ve).abs() <= ve.abs() * 1e-5 + 1e-6).all())


@pytest.mark.parametrize("C", [1, 2])
def test_workspace_contents_do_not_matter(te, C):
    """the planes of the pre-pass have columns nobody computes (halo columns of the statistics planes, alignment padding):
    a workspace full of NaN bit patterns gives the same volume / indices as one full of zeros"""
    from connecting_the_dots_amd import _lib
    L = _lib.lib()
    N, H, W, D = 2, 40, 260, 40
    rs = np.random.RandomState(77 + C)
    A = dev(rs.randn(N, C, H, W).astype(np.float32)); B = dev(rs.randn(C, H, W).astype(np.float32))
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for fill in (0x00, 0xFF):
        vol = torch.empty((N, D, H, W), device="cuda")
        if C == 1:
            ws = torch.full((L.ctd_xcorrvol_argmax_workspace_bytes(N, C, H, W, D, 9, 1),), fill, dtype=torch.uint8, device="cuda")
            idx = torch.empty((N, H, W), dtype=torch.int64, device="cuda"); best = torch.empty((N, H, W), device="cuda")
            assert L.ctd_xcorrvol_argmax_f32(A.data_ptr(), B.data_ptr(), 0, vol.data_ptr(), idx.data_ptr(), best.data_ptr(), N, C, H, W, D, 9,
                                             1, 1e-5, ws.data_ptr(), ws.numel(), 0, s) == 0
            outs.append((vol, idx, best))
        else:
            ws = torch.full((L.ctd_xcorrvol_workspace_bytes(N, C, H, W, D, 9, 1),), fill, dtype=torch.uint8, device="cuda")
            assert L.ctd_xcorrvol_f32(A.data_ptr(), B.data_ptr(), 0, vol.data_ptr(), N, C, H, W, D, 9, 1, ws.data_ptr(), ws.numel(), 0, s) == 0
            outs.append((vol,))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert bool(torch.isfinite(outs[1][0]).all())


def test_rank_single_disparity_with_listed_windows(te):
    """D = 1 with flat patches (found by tools/fuzz_rank.py): the lone candidate of a listed window is still re-scored when
    no volume is materialised -- the returned best score is the exact one, not the ranking's placeholder"""
    rs = np.random.RandomState(241)
    N, H, W = 3, 56, 396
    a = rs.randn(N, 1, H, W).astype(np.float32); b = rs.randn(1, H, W).astype(np.float32)
    b[:, 10:22, 40:54] = 0.3; b[:, : H // 2, :10] = 0.125; a[1, :, 30:40, 100:111] = -0.7
    check_both_modes(te, dev(a), dev(b), 1, "D = 1, flat patches")
