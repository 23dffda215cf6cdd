"""ctypes binding of libctd_hip.so (the C ABI of include/ctd_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libctd_hip.so")

_c_int, _c_long, _c_float, _c_size_t, _vp = (ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t,
                                             ctypes.c_void_p)

class PatternLevel(ctypes.Structure):
    """ctd_pattern_level of include/ctd_hip.h"""
    _fields_ = [("disp", _vp), ("im", _vp), ("mask", _vp), ("pattern", _vp), ("pattern_proj", _vp), ("grad_proj", _vp),
                ("grad_disp", _vp), ("B", _c_int), ("H", _c_int), ("W", _c_int)]


_levels_p = ctypes.POINTER(PatternLevel)

# name -> (restype, argtypes); mirrors include/ctd_hip.h one to one
SIGNATURES = {
    "ctd_version": (_c_int, []),
    "ctd_status_string": (ctypes.c_char_p, [_c_int]),
    "ctd_xcorrvol_workspace_bytes": (_c_size_t, [_c_int] * 7),
    "ctd_xcorrvol_f32": (_c_int, [_vp, _vp, _c_long, _vp] + [_c_int] * 7 + [_vp, _c_size_t, _c_int, _vp]),
    "ctd_xcorrvol_pattern_prepare_f32": (_c_int, [_vp, _c_long] + [_c_int] * 6 + [_vp, _c_size_t, _c_int, _vp]),
    "ctd_xcorrvol_f64": (_c_int, [_vp, _vp, _c_long, _vp] + [_c_int] * 6 + [_vp, _c_size_t, _c_int, _vp]),
    "ctd_argmax_disp_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 4 + [_c_int, _vp]),
    "ctd_xcorrvol_rank_supported": (_c_int, [_c_int] * 5),
    "ctd_xcorrvol_rank_layout": (_c_int, [_c_int] * 5 + [ctypes.POINTER(_c_size_t)]),
    "ctd_xcorrvol_argmax_workspace_bytes": (_c_size_t, [_c_int] * 7),
    "ctd_xcorrvol_argmax_f32": (_c_int, [_vp, _vp, _c_long, _vp, _vp, _vp] + [_c_int] * 7 + [_c_float, _vp, _c_size_t,
                                                                                          _c_int, _vp]),
    "ctd_lcn_xcorrvol_supported": (_c_int, [_c_int] * 5),
    "ctd_lcn_xcorrvol_argmax_f32": (_c_int, [_vp, _vp, _vp, _c_int, _c_float, _c_int, _vp, _c_long, _vp, _vp, _vp] + [_c_int] * 6 +
                                    [_c_float, _vp, _c_size_t, _c_int, _vp]),
    "ctd_photometric_fwd_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 6 + [_c_float, _c_int, _vp]),
    "ctd_photometric_fwd_f64": (_c_int, [_vp, _vp, _vp] + [_c_int] * 6 + [_c_float, _c_int, _vp]),
    "ctd_photometric_bwd_f32": (_c_int, [_vp, _vp, _vp, _vp] + [_c_int] * 6 + [_c_float, _c_int, _vp]),
    "ctd_photometric_bwd_f64": (_c_int, [_vp, _vp, _vp, _vp] + [_c_int] * 6 + [_c_float, _c_int, _vp]),
    "ctd_photometric_fwd_fast_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 6 + [_c_float, _c_int, _vp]),
    "ctd_photometric_bwd_fast_f32": (_c_int, [_vp, _vp, _vp, _vp] + [_c_int] * 6 + [_c_float, _c_int, _vp]),
    "ctd_costvol_f32": (_c_int, [_vp, _vp, _c_long, _vp] + [_c_int] * 6 + [_c_float, _c_int, _vp]),
    "ctd_costvol_workspace_bytes": (_c_size_t, [_c_int] * 7),
    "ctd_costvol_fast_f32": (_c_int, [_vp, _vp, _c_long, _vp] + [_c_int] * 6 + [_c_float, _vp, _c_size_t, _c_int, _vp]),
    "ctd_lcn_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 4 + [_c_float, _c_int, _vp]),
    "ctd_lcn_fast_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 4 + [_c_float, _c_int, _vp]),
    "ctd_lcn_datagen_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 4 + [_c_float, _c_int, _vp]),
    "ctd_disp_to_depth_fwd_f32": (_c_int, [_vp, _vp, _c_long, _c_float, _c_int, _vp]),
    "ctd_idx_to_depth_f32": (_c_int, [_vp, _vp, _c_long, _c_float, _c_float, _c_int, _vp]),
    "ctd_disp_to_depth_bwd_f32": (_c_int, [_vp, _vp, _vp, _c_long, _c_float, _c_int, _vp]),
    "ctd_disparity_loss_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "ctd_disparity_loss_fwd_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 3 + [_vp, _c_size_t, _c_int, _vp]),
    "ctd_disparity_loss_bwd_f32": (_c_int, [_vp] * 5 + [_c_int] * 3 + [_vp, _c_size_t, _c_int, _vp]),
    "ctd_geometric_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "ctd_geometric_fwd_f32": (_c_int, [_vp] * 9 + [_c_int] * 4 + [_c_float, _vp, _c_size_t, _c_int, _vp]),
    "ctd_geometric_sym_fwd_f32": (_c_int, [_vp] * 9 + [_c_int] * 3 + [_c_float, _vp, _c_size_t, _vp, _c_int, _vp]),
    "ctd_geometric_bwd_f32": (_c_int, [_vp] * 10 + [_c_int, _vp] + [_c_int] * 3 + [_c_float, _c_int, _vp]),
    "ctd_pattern_loss_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "ctd_pattern_loss_fwd_f32": (_c_int, [_vp] * 6 + [_c_int] * 4 + [_c_float, _vp, _c_size_t, _c_int, _vp]),
    "ctd_pattern_loss_bwd_f32": (_c_int, [_vp] * 8 + [_c_int] * 4 + [_c_float, _c_int, _vp]),
    "ctd_pattern_loss_multi_workspace_bytes": (_c_size_t, [_c_int, _levels_p]),
    "ctd_pattern_loss_multi_fwd_f32": (_c_int, [_c_int, _levels_p, _vp, _c_int, _c_float, _vp, _c_size_t, _c_int, _vp]),
    "ctd_pattern_loss_multi_bwd_f32": (_c_int, [_c_int, _levels_p, _vp, _vp, _c_int, _c_float, _c_int, _vp]),
    "ctd_render_mesh_proj_f32": (_c_int, [_vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp,
                                          _c_float, _c_float, _vp, _vp, _vp, _c_int, _vp]),
    "ctd_render_mesh_f32": (_c_int, [_vp, _vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp]),
    "ctd_nn_f32": (_c_int, [_vp, _vp, _c_long, _c_long, _vp, _c_int, _vp]),
    "ctd_nn_f64": (_c_int, [_vp, _vp, _c_long, _c_long, _vp, _c_int, _vp]),
    "ctd_crosscheck": (_c_int, [_vp, _vp, _c_long, _c_long, _vp, _c_int, _vp]),
    "ctd_proj_nn_f32": (_c_int, [_vp, _vp, _vp] + [_c_int] * 4 + [_vp, _c_int, _vp]),
    "ctd_proj_nn_f64": (_c_int, [_vp, _vp, _vp] + [_c_int] * 4 + [_vp, _c_int, _vp]),
}

# measurement hooks of include/ctd_hip_bench.h (bench.py, tools/): not part of the drop-in interface
BENCH_SIGNATURES = {
    "ctd_kernel_timing_enable": (None, [_c_int]),
    "ctd_kernel_timing_collect": (_c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_int)]),
}

_lib = None


def lib():
    """Loads the HIP library; raises if it has not been built (python -m connecting_the_dots_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "connecting_the_dots_amd: %s is missing -- build it with "
                "`python -m connecting_the_dots_amd.build` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(BENCH_SIGNATURES.items()):
            fn = getattr(l, name)       # AttributeError here means header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().ctd_status_string(status).decode()
        raise RuntimeError("%s failed: %s (status %d)" % (what, msg, status))
