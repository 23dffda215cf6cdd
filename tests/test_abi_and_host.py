"""CPU-only checks: the C-ABI library loads and exports every symbol include/ctd_hip.h declares, the ctypes
table matches the header, host-side argument validation works without a GPU, the Python surface mirrors the
reference's torchext names."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ctd_hip.h")
BENCH_HEADER = os.path.join(ROOT, "include", "ctd_hip_bench.h")


def declared_symbols(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from connecting_the_dots_amd import _lib
    from connecting_the_dots_amd import build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols() + declared_symbols(BENCH_HEADER)
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "libctd_hip.so does not export %s" % n


def test_ctypes_table_matches_header():
    from connecting_the_dots_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    # the measurement hooks live in a header of their own, outside the drop-in interface
    assert sorted(_lib.BENCH_SIGNATURES) == declared_symbols(BENCH_HEADER)
    assert not any("timing" in n for n in declared_symbols())
    assert _lib.lib().ctd_version() == 5
    assert _lib.lib().ctd_status_string(1) == b"invalid argument"


def test_workspace_query_and_validation_need_no_gpu():
    from connecting_the_dots_amd import _lib
    L = _lib.lib()
    assert L.ctd_xcorrvol_workspace_bytes(16, 1, 432, 512, 128, 9, 0) > 0
    assert L.ctd_xcorrvol_workspace_bytes(16, 1, 432, 512, 128, 9, 1) >= L.ctd_xcorrvol_workspace_bytes(16, 1, 432, 512, 128, 9, 0)
    assert L.ctd_xcorrvol_workspace_bytes(1, 1, 0, 512, 128, 9, 0) == 0
    # invalid sizes are rejected before any HIP call
    assert L.ctd_xcorrvol_f32(None, None, 0, None, 1, 1, -4, 8, 8, 9, 0, None, 0, -1, None) == 1
    assert L.ctd_lcn_f32(None, None, None, 1, 8, 8, 8, 0.05, -1, None) == 1          # radius >= H
    assert L.ctd_photometric_fwd_f32(None, None, None, 1, 1, 8, 8, 9, 7, 0.5, -1, None) == 1   # bad type
    # 2^31 outputs exceed the reference's int indexing (common_cuda.h:65-66)
    assert L.ctd_xcorrvol_f32(None, None, 0, None, 1, 1, 4096, 4096, 128, 9, 0, None, 0, -1, None) == 1


def test_python_surface_mirrors_reference_names():
    from connecting_the_dots_amd import torchext as te
    for name in ("nn", "NNFunction", "crosscheck", "CrossCheckFunction", "proj_nn", "ProjNNFunction", "xcorrvol",
                 "XCorrVolFunction", "photometric_loss", "PhotometricLossFunction", "photometric_loss_pytorch",
                 "CoordConv2d"):
        assert hasattr(te, name), name
    for name in ("xcorrvol_batch", "xcorrvol_argmax", "argmax_disp", "lcn", "LCN", "costvol", "disp_to_depth",
                 "disparity_loss", "geometric_loss", "DispToDepth", "DisparityLoss", "ProjectionDepthSimilarityLoss",
                 "RectifiedPatternSimilarityLoss"):
        assert hasattr(te, name), name


def test_cpu_tensors_are_rejected_loudly():
    from connecting_the_dots_amd import torchext as te
    x = torch.rand(1, 8, 8)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        te.xcorrvol(x, x, 4, 9)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        te.photometric_loss(x[None], x[None], 9)
    with pytest.raises(Exception, match="invalid loss type"):
        te.photometric_loss(x[None], x[None], 9, "nope")


def test_photometric_pytorch_formulation_matches_oracle(oracle):
    """photometric_loss_pytorch runs on CPU tensors (stock ops) and agrees with the oracle."""
    import numpy as np
    from connecting_the_dots_amd import torchext as te
    from tests.util import assert_close
    rs = np.random.RandomState(3)
    es, ta = rs.rand(1, 2, 12, 14), rs.rand(1, 2, 12, 14)
    for ty, name in enumerate(("mse", "sad", "census_mse", "census_sad")):
        got = te.photometric_loss_pytorch(torch.from_numpy(es), torch.from_numpy(ta), 9, name, 0.5).numpy()
        assert_close(got, oracle.photometric_fwd(es, ta, 9, ty, 0.5), rtol=1e-10, atol=1e-12, what=name)


def test_coordconv_grid():
    from connecting_the_dots_amd import torchext as te
    m = te.CoordConv2d(1, 2, 3, 1, 1)
    y = m(torch.zeros(2, 1, 5, 7))
    assert y.shape == (2, 2, 5, 7)
    assert m.uv.shape == (1, 2, 5, 7)
    assert float(m.uv[0, 0, 0, 0]) == -1 and float(m.uv[0, 0, 0, -1]) == 1       # u spans [-1, 1] (modules.py:19)
    assert float(m.uv[0, 1, 0, 0]) == -1 and float(m.uv[0, 1, -1, 0]) == 1


def test_all_d_plan_of_the_benchmark_shapes():
    """Host logic of the ranked argmax (no GPU work): how the all-D kernel cuts the BASELINE shapes -- band height and
    passes over the disparities (ctd_xcorrvol_rank_layout offsets[0] / [4]) -- and that its workspace query covers it."""
    import ctypes
    from connecting_the_dots_amd import _lib
    L = _lib.lib()
    off = (ctypes.c_size_t * 5)()
    assert L.ctd_xcorrvol_rank_layout(16, 432, 512, 128, 0, off) == 0
    assert (off[0], off[4]) == (54, 5)             # config 2: one round of 256 workgroups, 5 passes of 26 disparities
    assert L.ctd_xcorrvol_rank_layout(1, 1024, 1024, 256, 0, off) == 0
    assert (off[0], off[4]) == (16, 9)             # config 4: 64 bands x 4 tiles = 256 workgroups, 9 passes
    assert L.ctd_xcorrvol_rank_layout(1, 9, 4, 1, 0, off) == 0 and 1 <= off[0] <= 9 and off[4] == 1
    assert L.ctd_xcorrvol_rank_supported(1, 432, 512, 128, 9) == 1
    assert L.ctd_xcorrvol_rank_supported(1, 432, 510, 128, 9) == 0      # W % 4
    assert L.ctd_xcorrvol_rank_supported(1, 432, 512, 600, 9) == 0      # D > 512
    need = L.ctd_xcorrvol_argmax_workspace_bytes(16, 1, 432, 512, 128, 9, 1)
    assert need >= L.ctd_xcorrvol_workspace_bytes(16, 1, 432, 512, 128, 9, 1) > 0


def test_ticket_words_are_cleared_when_a_launch_returns_an_error():
    """A launch that takes its ticket words at zero and fails must not leave them dirty for the next call."""
    from connecting_the_dots_amd.torchext import functions as F
    ticket = torch.zeros(128, dtype=torch.int32)

    def failing(tk):
        tk[3] = 7                                            # what an aborted launch may leave behind
        return 2                                             # CTD_ERR_WORKSPACE
    with pytest.raises(RuntimeError, match="workspace"):
        F._call_with_ticket(failing, ticket, "geometric_loss")
    assert int(ticket.abs().sum()) == 0
    F._call_with_ticket(lambda tk: 0, ticket, "geometric_loss")   # a good call passes straight through


def test_kernel_timing_state_is_per_thread():
    """The measurement hooks keep their state per calling thread: enabling them in one thread does not instrument (or
    race with) launches of another.  Without a GPU: collect() in a thread that never enabled them reports nothing."""
    import threading
    from connecting_the_dots_amd import _lib
    L = _lib.lib()
    seen = {}

    def other():
        seen["n"] = L.ctd_kernel_timing_collect(None, None)
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["n"] == 0
