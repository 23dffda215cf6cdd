"""ctypes front-end of the CPU oracle (libctd_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module.  All functions take and return numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_c_int, _c_long, _c_float, _vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p


def build(force=False):
    so = os.path.join(HERE, "libctd_oracle.so")
    srcs = [os.path.join(HERE, f) for f in ("ctd_oracle.c", "ctd_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, "-B", "libctd_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _sfx(a):
    if a.dtype == np.float32:
        return "f32"
    if a.dtype == np.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64, got %s" % a.dtype)


def _p(a):
    return a.ctypes.data_as(_vp)


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


def xcorrvol(in0, in1, n_disps, block_size, nthreads=1):
    """[C,H,W] x [C,H,W] -> [D,H,W]   (torchext/ext/ext.h:120-191)"""
    in0, in1 = _c(in0), _c(in1, in0.dtype)
    C, H, W = in0.shape
    out = np.empty((n_disps, H, W), in0.dtype)
    fn = getattr(lib(), "ctd_oracle_xcorrvol_" + _sfx(in0))
    fn.argtypes = [_vp, _vp, _vp, _c_long, _c_long, _c_long, _c_long, _c_long, _c_int]
    rc = fn(_p(in0), _p(in1), _p(out), C, H, W, n_disps, block_size, nthreads)
    assert rc == 0
    return out


def argmax(vol):
    """[D,H,W] -> (idx int64 [H,W], best [H,W]); first index wins ties."""
    vol = _c(vol)
    D, H, W = vol.shape
    idx = np.empty((H, W), np.int64)
    best = np.empty((H, W), vol.dtype)
    fn = getattr(lib(), "ctd_oracle_argmax_" + _sfx(vol))
    fn.argtypes = [_vp, _vp, _vp, _c_long, _c_long, _c_long]
    assert fn(_p(vol), _p(idx), _p(best), D, H, W) == 0
    return idx, best


def photometric_fwd(es, ta, block_size, type, eps, nthreads=1):
    """[B,C,H,W] x2 -> [B,1,H,W]   (ext.h:201-266)"""
    es, ta = _c(es), _c(ta, es.dtype)
    B, C, H, W = es.shape
    out = np.empty((B, 1, H, W), es.dtype)
    fn = getattr(lib(), "ctd_oracle_photometric_fwd_" + _sfx(es))
    fn.argtypes = [_vp, _vp, _vp] + [_c_int] * 6 + [_c_float, _c_int]
    rc = fn(_p(es), _p(ta), _p(out), B, C, H, W, block_size, type, eps, nthreads)
    assert rc == 0
    return out


def photometric_bwd(es, ta, grad_out, block_size, type, eps):
    """-> grad wrt es [B,C,H,W]   (ext.h:268-344), serial scatter order."""
    es, ta, grad_out = _c(es), _c(ta, es.dtype), _c(grad_out, es.dtype)
    B, C, H, W = es.shape
    gi = np.empty((B, C, H, W), es.dtype)
    fn = getattr(lib(), "ctd_oracle_photometric_bwd_" + _sfx(es))
    fn.argtypes = [_vp, _vp, _vp, _vp] + [_c_int] * 6 + [_c_float]
    rc = fn(_p(es), _p(ta), _p(grad_out), _p(gi), B, C, H, W, block_size, type, eps)
    assert rc == 0
    return gi


def costvol(im, pattern, n_disps, block_size, type, eps, nthreads=1):
    """SAD / census cost volume by composition:
    cost[d] = photometric_fwd(es=P_d, ta=I)[0,0],  P_d[h,x] = P[h, clamp(x-d)].
    im, pattern [H,W] -> [D,H,W]."""
    im, pattern = _c(im), _c(pattern, im.dtype)
    H, W = im.shape
    out = np.empty((n_disps, H, W), im.dtype)
    cols = np.arange(W)
    for d in range(n_disps):
        pd = pattern[:, np.clip(cols - d, 0, W - 1)]
        out[d] = photometric_fwd(pd[None, None], im[None, None], block_size, type, eps, nthreads)[0, 0]
    return out


def lcn(x, radius, eps):
    """networks.LCN.tforward (model/networks.py:523-533). x [N,1,H,W] f32."""
    x = _c(x, np.float32)
    N, one, H, W = x.shape
    assert one == 1
    y = np.empty_like(x)
    s = np.empty_like(x)
    fn = lib().ctd_oracle_lcn_f32
    fn.argtypes = [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float]
    assert fn(_p(x), _p(y), _p(s), N, H, W, radius, eps) == 0
    return y, s


def lcn_datagen(img, ks, eps):
    """data/lcn/lcn.pyx:16-58. img [H,W] f32 -> (lcn, raw std)."""
    img = _c(img, np.float32)
    H, W = img.shape
    y = np.empty_like(img)
    s = np.empty_like(img)
    fn = lib().ctd_oracle_lcn_datagen_f32
    fn.argtypes = [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_float]
    assert fn(_p(img), _p(y), _p(s), H, W, ks, eps) == 0
    return y, s


def disp_to_depth(disp, bf):
    """networks.DispToDepth.tforward (model/networks.py:318-321)."""
    disp = _c(disp, np.float32)
    out = np.empty_like(disp)
    fn = lib().ctd_oracle_disp_to_depth_f32
    fn.argtypes = [_vp, _vp, _c_long, _c_float]
    assert fn(_p(disp), _p(out), disp.size, bf) == 0
    return out


def nn(in0, in1):
    """[n0,3] x [n1,3] -> int64 [n0]   (ext.h:13-47)"""
    in0, in1 = _c(in0), _c(in1, in0.dtype)
    out = np.empty((in0.shape[0],), np.int64)
    fn = getattr(lib(), "ctd_oracle_nn_" + _sfx(in0))
    fn.argtypes = [_vp, _vp, _c_long, _c_long, _vp]
    assert fn(_p(in0), _p(in1), in0.shape[0], in1.shape[0], _p(out)) == 0
    return out


def crosscheck(in0, in1):
    """int64 [n0] x int64 [n1] -> uint8 [n0]   (ext.h:49-66)"""
    in0, in1 = _c(in0, np.int64), _c(in1, np.int64)
    out = np.empty((in0.shape[0],), np.uint8)
    fn = lib().ctd_oracle_crosscheck
    fn.argtypes = [_vp, _vp, _c_long, _c_long, _vp]
    assert fn(_p(in0), _p(in1), in0.shape[0], in1.shape[0], _p(out)) == 0
    return out


def proj_nn(xyz0, xyz1, K, patch_size):
    """[B,H,W,3] x2, K [3,3] -> int64 [B,H,W]   (ext.h:68-117)"""
    xyz0 = _c(xyz0)
    xyz1, K = _c(xyz1, xyz0.dtype), _c(K, xyz0.dtype)
    B, H, W, _ = xyz0.shape
    out = np.empty((B, H, W), np.int64)
    fn = getattr(lib(), "ctd_oracle_proj_nn_" + _sfx(xyz0))
    fn.argtypes = [_vp, _vp, _vp, _c_long, _c_long, _c_long, _c_long, _vp]
    assert fn(_p(xyz0), _p(xyz1), _p(K), B, H, W, patch_size, _p(out)) == 0
    return out


def _cam_params(K, R, t):
    """{fx, fy, px, py, R[9], t[3]} as the reference's Camera<T> constructor takes them (render.h:24)"""
    return np.ascontiguousarray(np.concatenate([[K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.asarray(R).reshape(9),
                                                np.asarray(t).reshape(3)]).astype(np.float32))


def render_mesh(verts, colors, normals, faces, cam, shader, nthreads=1, fn=None, **_unused):
    """RenderMeshFunctor (renderer/render/render.h:150-223).  cam = (K, R, t, width, height) -> depth [H,W],
    color [H,W,3], normal [H,W,3].  `fn`: the reference harness's entry point (same signature)."""
    verts, colors, normals = _c(verts, np.float32), _c(colors, np.float32), _c(normals, np.float32)
    faces = _c(faces, np.int32)
    cp = _cam_params(*cam[:3])
    W, H = int(cam[3]), int(cam[4])
    depth = np.zeros((H, W), np.float32)
    color = np.zeros((H, W, 3), np.float32)
    normal = np.zeros((H, W, 3), np.float32)
    sh = np.asarray(shader, np.float32)
    if fn is None:
        fn = lib().ctd_oracle_render_mesh_f32
    fn.argtypes = [_vp, _vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int]
    rc = fn(_p(verts), _p(colors), _p(normals), verts.shape[0], _p(faces), faces.shape[0], _p(cp), W, H, _p(sh), _p(depth),
            _p(color), _p(normal), nthreads)
    assert rc == 0
    return depth, color, normal


def render_mesh_proj(verts, colors, faces, cam, proj, shader, pattern, d_alpha, d_beta, nthreads=1, fn=None):
    """RenderProjectorFunctor (renderer/render/render.h:251-364).  cam / proj = (K [3,3], R [3,3], t [3], width,
    height); shader = (ka, kd, ks, alpha); pattern [ph, pw, 3] -> depth [H,W], color [H,W,3], normal [H,W,3]
    (normal zero-initialised here; the reference leaves it untouched where nothing is hit)."""
    verts, colors = _c(verts, np.float32), _c(colors, np.float32)
    faces = _c(faces, np.int32)
    pattern = _c(pattern, np.float32)
    cp, pp = _cam_params(*cam[:3]), _cam_params(*proj[:3])
    W, H = int(cam[3]), int(cam[4])
    assert pattern.shape == (int(proj[4]), int(proj[3]), 3)
    depth = np.zeros((H, W), np.float32)
    color = np.zeros((H, W, 3), np.float32)
    normal = np.zeros((H, W, 3), np.float32)
    sh = np.asarray(shader, np.float32)
    if fn is None:
        fn = lib().ctd_oracle_render_mesh_proj_f32
        fn.argtypes = [_vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_float,
                       _c_float, _vp, _vp, _vp, _c_int]
        rc = fn(_p(verts), _p(colors), verts.shape[0], _p(faces), faces.shape[0], _p(cp), W, H, _p(pp), int(proj[3]),
                int(proj[4]), _p(sh), _p(pattern), d_alpha, d_beta, _p(depth), _p(color), _p(normal), nthreads)
    else:                                   # the reference harness (oracle/ref_render_driver.cpp) takes normals too
        fn.argtypes = [_vp, _vp, _vp, _c_int, _vp, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_float,
                       _c_float, _vp, _vp, _vp, _c_int]
        rc = fn(_p(verts), _p(colors), None, verts.shape[0], _p(faces), faces.shape[0], _p(cp), W, H, _p(pp),
                int(proj[3]), int(proj[4]), _p(sh), _p(pattern), d_alpha, d_beta, _p(depth), _p(color), _p(normal), nthreads)
    assert rc == 0
    return depth, color, normal
