"""Mirror of the reference's `torchext/functions.py` on top of libctd_hip.so.

Same names, argument meaning, autograd contract and error behaviour as the reference
(torchext/functions.py:1-147); every op takes CUDA(=HIP) tensors that are contiguous and
launches on the caller's current stream and the tensor's device.  There is no CPU path in
this package: CPU tensors raise (the reference's CPU path is `ext_cpu`, which lives on as
the test oracle only).

Additive API (no reference counterpart): `xcorrvol_batch`, `argmax_disp`,
`xcorrvol_argmax`, `lcn`, ... and the keyword `algo` of the ops that have two kernel families:
    'fast'  (default) tolerance-level kernels, |a-b| <= 1e-5*|b| + 1e-6 against the reference (LDS-tiled, HBM- or
            issue-bound; f32 and odd block sizes up to 9 -- anything else runs the reference-order kernels);
    'exact' the reference's operation order, bit-identical to its CPU build (the parity anchor).
The default can be changed with the environment variables CTD_NCC_ALGO (xcorrvol family) and CTD_PHOTO_ALGO
(photometric loss, cost volumes, pattern similarity loss).
"""
import os

import torch

from .. import _lib

_ALGOS = {"exact": 0, "fast": 1}


def _default_algo():
    return os.environ.get("CTD_NCC_ALGO", "fast")


def _ncc_fast_covers(dtype, n_disps, block_size):
    """the separable-sum kernels: f32, block 3/5/7/9, D <= 512; everything else is the reference-order kernel's"""
    return dtype == torch.float32 and int(block_size) in (3, 5, 7, 9) and int(n_disps) <= 512


def _check(t, name, dtypes=(torch.float32, torch.float64)):
    # CHECK_CUDA / CHECK_CONTIGUOUS of the reference binding (ext.h:6-10) -> RuntimeError
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor (connecting_the_dots_amd has no CPU path)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype not in dtypes:
        raise RuntimeError("%s: unsupported dtype %s" % (name, t.dtype))


def _same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device (%s vs %s)" % (dev, t.device))
    return dev


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _workspace(nbytes, dev):
    # torch's caching allocator makes this a stream-ordered sub-allocation, no hipMalloc
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def _ptr(t):
    return t.data_ptr() if t is not None else None


_TICKETS = {}


def _ticket(dev):
    """Zeroed device words (128 of them) per (device, stream) for kernels whose last workgroup finishes a reduction: they
    take them at zero and leave them at zero (ctd_hip.h: ctd_geometric_sym_fwd_f32)."""
    key = (dev.index, _stream(dev))
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(128, dtype=torch.int32, device=dev)
    return t


def _call_with_ticket(call, ticket, what):
    """Runs `call(ticket)` (a C-ABI launch that takes its ticket words at zero and leaves them at zero).  When it
    returns an error the words may be anywhere in between -- a later launch on them would then have no "last"
    workgroup and leave its result unwritten -- so they are cleared (stream-ordered, like the launch) before the error
    is raised."""
    st = call(ticket)
    if st != 0:
        ticket.zero_()
    _lib.check(st, what)


# --------------------------------------------------------------------------------------
# Nearest-neighbour consistency ops (reference: NNFunction / CrossCheckFunction / ProjNNFunction,
# functions.py:5-56; bindings ext_cuda.cpp:17-68)
# --------------------------------------------------------------------------------------
class NNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in0, in1):
        _check(in0, "in0")
        _check(in1, "in1")
        dev = _same_device(in0, in1)
        # shape asserts of the reference binding (ext_cuda.cpp:25-28)
        if in0.dim() != 2 or in1.dim() != 2:
            raise RuntimeError("in0 has to be N0 x 3, in1 has to be N1 x 3")
        if in0.shape[1] != in1.shape[1]:
            raise RuntimeError("in0 and in1 have to be the same shape")
        if in0.shape[1] != 3:
            raise RuntimeError("dim hast to be 3")
        if in0.dtype != in1.dtype:
            raise RuntimeError("in0 and in1 must have the same dtype")
        out = torch.empty((in0.shape[0],), dtype=torch.int64, device=dev)
        fn = _lib.lib().ctd_nn_f32 if in0.dtype == torch.float32 else _lib.lib().ctd_nn_f64
        _lib.check(fn(_ptr(in0), _ptr(in1), in0.shape[0], in1.shape[0], _ptr(out), dev.index, _stream(dev)), "nn")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return None, None


def nn(in0, in1):
    """Index of the nearest in1 point [N1,3] for every in0 point [N0,3] (int64 [N0], -1 if none within 1e9)."""
    return NNFunction.apply(in0, in1)


class CrossCheckFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in0, in1):
        _check(in0, "in0", (torch.int64,))
        _check(in1, "in1", (torch.int64,))
        dev = _same_device(in0, in1)
        if in0.dim() != 1 or in1.dim() != 1:
            raise RuntimeError("crosscheck expects 1-D index tensors")
        out = torch.empty((in0.shape[0],), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().ctd_crosscheck(_ptr(in0), _ptr(in1), in0.shape[0], in1.shape[0], _ptr(out), dev.index,
                                             _stream(dev)), "crosscheck")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return None, None


def crosscheck(in0, in1):
    """uint8 [N0]: 1 where in1[in0[i]] == i (mutual nearest neighbours)."""
    return CrossCheckFunction.apply(in0, in1)


class ProjNNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz0, xyz1, K, patch_size):
        _check(xyz0, "xyz0")
        _check(xyz1, "xyz1")
        _check(K, "K")
        dev = _same_device(xyz0, xyz1, K)
        if xyz0.dim() != 4 or xyz1.dim() != 4 or xyz0.shape[3] != 3 or tuple(xyz0.shape) != tuple(xyz1.shape):
            raise RuntimeError("proj_nn expects xyz0 and xyz1 of the same shape [B,H,W,3]")
        if K.numel() != 9:
            raise RuntimeError("K has to be 3 x 3")
        if not (xyz0.dtype == xyz1.dtype == K.dtype):
            raise RuntimeError("xyz0, xyz1 and K must have the same dtype")
        B, H, W, _ = xyz0.shape
        out = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        fn = _lib.lib().ctd_proj_nn_f32 if xyz0.dtype == torch.float32 else _lib.lib().ctd_proj_nn_f64
        _lib.check(fn(_ptr(xyz0), _ptr(xyz1), _ptr(K), B, H, W, int(patch_size), _ptr(out), dev.index, _stream(dev)),
                   "proj_nn")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return None, None, None, None


def proj_nn(xyz0, xyz1, K, patch_size):
    """Flat index (int64 [B,H,W]) of the xyz1 point closest to each xyz0 point inside the patch_size^2 patch
    around its projection with K; -1 where the patch falls outside the image."""
    return ProjNNFunction.apply(xyz0, xyz1, K, patch_size)


# --------------------------------------------------------------------------------------
# NCC volume (reference: XCorrVolFunction, functions.py:59-74)
# --------------------------------------------------------------------------------------
def _xcorrvol_impl(in0, in1, n_disps, block_size, algo, prepared=None):
    """in0 [N,C,H,W], in1 [C,H,W] | [N,C,H,W] -> [N,D,H,W]"""
    L = _lib.lib()
    dev = _same_device(in0, in1)
    if in0.dtype != in1.dtype:
        raise RuntimeError("in0 and in1 must have the same dtype")
    N, C, H, W = in0.shape
    if tuple(in1.shape[-3:]) != (C, H, W):
        raise RuntimeError("in0 and in1 must have the same [C,H,W] shape")
    stride1 = 0 if in1.dim() == 3 else C * H * W
    if in1.dim() == 4 and in1.shape[0] != N:
        raise RuntimeError("in1 batch does not match in0")
    D, bs = int(n_disps), int(block_size)
    out = torch.empty((N, D, H, W), dtype=in0.dtype, device=dev)
    if algo not in _ALGOS:
        raise RuntimeError("unknown algo %r" % (algo,))
    if algo == "fast" and not _ncc_fast_covers(in0.dtype, D, bs):
        algo = "exact"
    a = _ALGOS[algo]
    if prepared is not None:
        _check_prepared(prepared, "xcorrvol_batch", a, in1, N, D, bs, dev)
        ws, a = prepared.workspace, a | 0x100                        # CTD_PATTERN_PREPARED
    else:
        ws = _workspace(L.ctd_xcorrvol_workspace_bytes(N, C, H, W, D, bs, a), dev)
    if in0.dtype == torch.float32:
        st = L.ctd_xcorrvol_f32(_ptr(in0), _ptr(in1), stride1, _ptr(out), N, C, H, W, D, bs, a, _ptr(ws),
                                ws.numel(), dev.index, _stream(dev))
    else:
        st = L.ctd_xcorrvol_f64(_ptr(in0), _ptr(in1), stride1, _ptr(out), N, C, H, W, D, bs, _ptr(ws), ws.numel(),
                                dev.index, _stream(dev))
    _lib.check(st, "xcorrvol")
    return out


class XCorrVolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in0, in1, n_disps, block_size, algo=None):
        _check(in0, "in0")
        _check(in1, "in1")
        if in0.dim() != 3 or in1.dim() != 3:
            raise RuntimeError("xcorrvol expects [C,H,W] tensors")
        return _xcorrvol_impl(in0.unsqueeze(0), in1, n_disps, block_size, algo or _default_algo())[0]

    @staticmethod
    def backward(ctx, grad_out):
        return None, None, None, None, None


def xcorrvol(in0, in1, n_disps, block_size, algo=None):
    """Zero-mean NCC volume [D,H,W] between in0 [C,H,W] at (h,w) and in1 at (h,w-d).
    algo (additive): 'fast' (default) | 'exact', see the module docstring."""
    return XCorrVolFunction.apply(in0, in1, n_disps, block_size, algo)


def xcorrvol_batch(in0, in1, n_disps, block_size, algo=None, prepared=None):
    """Additive: xcorrvol for a batch of frames in one launch.
    in0 [N,C,H,W]; in1 [C,H,W] (shared pattern) or [N,C,H,W] -> [N,D,H,W].  `prepared`: see `prepare_pattern`."""
    _check(in0, "in0")
    _check(in1, "in1")
    if in0.dim() != 4 or in1.dim() not in (3, 4):
        raise RuntimeError("xcorrvol_batch expects in0 [N,C,H,W] and in1 [C,H,W] or [N,C,H,W]")
    return _xcorrvol_impl(in0, in1, n_disps, block_size, algo or _default_algo(), prepared)


def argmax_disp(vol):
    """Additive: (torch.argmax(vol, -3), vol.max(-3)) for vol [D,H,W] or [N,D,H,W]; first index wins ties."""
    _check(vol, "vol", (torch.float32,))
    squeeze = vol.dim() == 3
    v = vol.unsqueeze(0) if squeeze else vol
    if v.dim() != 4:
        raise RuntimeError("argmax_disp expects [D,H,W] or [N,D,H,W]")
    N, D, H, W = v.shape
    dev = v.device
    idx = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    best = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    st = _lib.lib().ctd_argmax_disp_f32(_ptr(v), _ptr(idx), _ptr(best), N, D, H, W, dev.index, _stream(dev))
    _lib.check(st, "argmax_disp")
    return (idx[0], best[0]) if squeeze else (idx, best)


def _check_prepared(prepared, who, a, in1, N, D, bs, dev):
    if (a != 1 or prepared.in1 is not in1 or prepared.n_frames != N or prepared.n_disps != D or prepared.block_size != bs
            or prepared.workspace.device != dev):
        raise RuntimeError("%s: `prepared` belongs to another pattern, frame count, shape or device "
                           "(or the call does not take the fast path)" % who)


class PreparedPattern:
    """The pattern half of the fast NCC path, done once (`prepare_pattern`): holds the workspace the pattern's window
    statistics and fix-up lists live in -- dedicated to the calls that pass this object -- and the shape it is valid for."""

    def __init__(self, in1, n_frames, n_disps, block_size, workspace):
        self.in1, self.n_frames, self.n_disps, self.block_size, self.workspace = in1, n_frames, n_disps, block_size, workspace


def prepare_pattern(in1, n_frames, n_disps, block_size):
    """Additive: window statistics / fix-up lists of the pattern `in1` ([1,H,W] shared, or [N,1,H,W] per frame), computed
    ONCE for all later `xcorrvol_argmax(..., prepared=handle)` calls with `n_frames` frames of the same shape -- the
    reference prepares its pattern once per run too (model/exp_synph.py:64-71).  Fast path, C == 1."""
    _check(in1, "in1", (torch.float32,))
    if in1.dim() not in (3, 4) or in1.shape[-3] != 1:
        raise RuntimeError("prepare_pattern expects in1 [1,H,W] or [N,1,H,W]")
    H, W = in1.shape[-2:]
    D, bs, N = int(n_disps), int(block_size), int(n_frames)
    if not _ncc_fast_covers(in1.dtype, D, bs):
        raise RuntimeError("prepare_pattern: the fast NCC path does not cover this block size / disparity range")
    L = _lib.lib()
    dev = in1.device
    ws = _workspace(L.ctd_xcorrvol_argmax_workspace_bytes(N, 1, H, W, D, bs, 1), dev)
    stride1 = 0 if in1.dim() == 3 else H * W
    st = L.ctd_xcorrvol_pattern_prepare_f32(_ptr(in1), stride1, N, 1, H, W, D, bs, _ptr(ws), ws.numel(), dev.index, _stream(dev))
    _lib.check(st, "prepare_pattern")
    return PreparedPattern(in1, N, D, bs, ws)


def xcorrvol_argmax(in0, in1, n_disps, block_size, return_volume=False, algo=None, rerank_eps=1e-5, prepared=None):
    """Additive: fused NCC volume + argmax over disparity (C == 1).
    in0 [N,1,H,W] | [1,H,W]; in1 [1,H,W] | [N,1,H,W].
    Returns (idx int64, best f32[, volume]); idx == torch.argmax(xcorrvol(...), 0) of the reference.
    `prepared`: a `prepare_pattern` handle of the same `in1`, frame count and shape (fast path only): the pattern half of
    the pre-pass is skipped and the handle's workspace is used."""
    _check(in0, "in0", (torch.float32,))
    _check(in1, "in1", (torch.float32,))
    squeeze = in0.dim() == 3
    a0 = in0.unsqueeze(0) if squeeze else in0
    if a0.dim() != 4 or in1.dim() not in (3, 4):
        raise RuntimeError("xcorrvol_argmax expects in0 [N,1,H,W] or [1,H,W]")
    L = _lib.lib()
    dev = _same_device(a0, in1)
    N, C, H, W = a0.shape
    if tuple(in1.shape[-3:]) != (C, H, W):
        raise RuntimeError("in0 and in1 must have the same [C,H,W] shape")
    stride1 = 0 if in1.dim() == 3 else C * H * W
    D, bs = int(n_disps), int(block_size)
    algo = algo or _default_algo()
    if algo not in _ALGOS:
        raise RuntimeError("unknown algo %r" % (algo,))
    if algo == "fast" and not _ncc_fast_covers(a0.dtype, D, bs):
        algo = "exact"
    a = _ALGOS[algo]
    idx = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    best = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    # the fast path ranks inside the volume kernel where it can (no volume needed); other shapes rank a materialised one
    need_vol = a == 1 and not L.ctd_xcorrvol_rank_supported(C, H, W, D, bs)
    vol = torch.empty((N, D, H, W), dtype=torch.float32, device=dev) if (return_volume or need_vol) else None
    if prepared is not None:
        _check_prepared(prepared, "xcorrvol_argmax", a, in1, N, D, bs, dev)
        ws, a_flag = prepared.workspace, a | 0x100                   # CTD_PATTERN_PREPARED
    else:
        ws, a_flag = _workspace(L.ctd_xcorrvol_argmax_workspace_bytes(N, C, H, W, D, bs, a), dev), a
    st = L.ctd_xcorrvol_argmax_f32(_ptr(a0), _ptr(in1), stride1, _ptr(vol), _ptr(idx), _ptr(best), N, C, H, W, D, bs,
                                   a_flag, float(rerank_eps), _ptr(ws), ws.numel(), dev.index, _stream(dev))
    _lib.check(st, "xcorrvol_argmax")
    if squeeze:
        idx, best = idx[0], best[0]
        vol = vol[0] if vol is not None else None
    return (idx, best, vol) if return_volume else (idx, best)


# --------------------------------------------------------------------------------------
# LCN (reference: networks.LCN.tforward, model/networks.py:523-533 -- not part of the
# reference's torchext; exposed here so the whole pre-normalisation is one kernel)
# --------------------------------------------------------------------------------------
def lcn(data, radius, epsilon, algo=None):
    """data [N,1,H,W] f32 -> ((data - avg) / std, std), std = sqrt(E[x^2] - avg^2 + 1e-6) + epsilon,
    box statistics over a (2*radius+1)^2 reflect-padded window.  Not differentiable (the reference
    only ever applies it to input images, exp_synph.py:84-91).
    algo (additive): 'exact' (default: f64 box sums, bit-identical to the oracle) | 'fast' (radius 1 .. 7, other radii run 'exact': f32 sliding sums of
    tile-centred samples, within 1e-5 |b| + 1e-6 of 'exact' and of the reference -- whose own summation order is unspecified --
    except on the variance floor of flat non-zero levels, see include/ctd_hip.h)."""
    _check(data, "data", (torch.float32,))
    if data.dim() != 4 or data.shape[1] != 1:
        raise RuntimeError("lcn expects [N,1,H,W]")
    N, _, H, W = data.shape
    if not (0 <= int(radius) < min(H, W)):
        raise RuntimeError("lcn: radius must be smaller than the image (ReflectionPad2d rule)")
    dev = data.device
    y = torch.empty_like(data)
    std = torch.empty_like(data)
    algo = algo or "exact"
    if algo not in ("exact", "fast"):
        raise RuntimeError("unknown algo %r" % (algo,))
    fn = _lib.lib().ctd_lcn_fast_f32 if (algo == "fast" and 1 <= int(radius) <= 7) else _lib.lib().ctd_lcn_f32
    st = fn(_ptr(data), _ptr(y), _ptr(std), N, H, W, int(radius), float(epsilon), dev.index, _stream(dev))
    _lib.check(st, "lcn")
    return y, std


def lcn_xcorrvol_argmax(raw, in1, n_disps, block_size, radius=5, epsilon=0.05, return_volume=False, lcn_algo="exact",
                        rerank_eps=1e-5, prepared=None):
    """Additive: `lcn` of the raw frames, then `xcorrvol_argmax(..., algo='fast')` against the (already LCN'd) pattern,
    as ONE call whose first kernel streams the raw frames once and leaves both the LCN outputs and the matcher's window
    statistics (ctd_lcn_xcorrvol_argmax_f32).  raw [N,1,H,W]; in1 [1,H,W] | [N,1,H,W].
    Returns (lcn, std, idx, best[, volume]).  lcn_algo 'exact': lcn / std carry the bits of `lcn(..., algo='exact')`;
    'fast': f32 box sums, tolerance level on well-conditioned windows only (see include/ctd_hip.h).
    Shapes the fused kernel does not cover run the two calls it replaces."""
    _check(raw, "raw", (torch.float32,))
    _check(in1, "in1", (torch.float32,))
    if raw.dim() != 4 or raw.shape[1] != 1 or in1.dim() not in (3, 4):
        raise RuntimeError("lcn_xcorrvol_argmax expects raw [N,1,H,W] and in1 [1,H,W] or [N,1,H,W]")
    if lcn_algo not in ("exact", "fast"):
        raise RuntimeError("unknown lcn_algo %r" % (lcn_algo,))
    L = _lib.lib()
    dev = _same_device(raw, in1)
    N, C, H, W = raw.shape
    if tuple(in1.shape[-3:]) != (C, H, W):
        raise RuntimeError("raw and in1 must have the same [C,H,W] shape")
    D, bs = int(n_disps), int(block_size)
    if not (0 <= int(radius) < min(H, W)):
        raise RuntimeError("lcn: radius must be smaller than the image (ReflectionPad2d rule)")
    if not L.ctd_lcn_xcorrvol_supported(H, W, D, int(radius), bs):
        y, std = lcn(raw, radius, epsilon, algo=lcn_algo)
        out = xcorrvol_argmax(y, in1, D, bs, return_volume=return_volume, algo="fast", rerank_eps=rerank_eps, prepared=prepared)
        return (y, std) + tuple(out)
    stride1 = 0 if in1.dim() == 3 else C * H * W
    y, std = torch.empty_like(raw), torch.empty_like(raw)
    idx = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    best = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    vol = torch.empty((N, D, H, W), dtype=torch.float32, device=dev) if return_volume else None
    a = 1
    if prepared is not None:
        _check_prepared(prepared, "lcn_xcorrvol_argmax", a, in1, N, D, bs, dev)
        ws, a = prepared.workspace, a | 0x100                        # CTD_PATTERN_PREPARED
    else:
        ws = _workspace(L.ctd_xcorrvol_argmax_workspace_bytes(N, C, H, W, D, bs, a), dev)
    st = L.ctd_lcn_xcorrvol_argmax_f32(_ptr(raw), _ptr(y), _ptr(std), int(radius), float(epsilon),
                                       0 if lcn_algo == "exact" else 1, _ptr(in1), stride1, _ptr(vol), _ptr(idx), _ptr(best),
                                       N, H, W, D, bs, a, float(rerank_eps), _ptr(ws), ws.numel(), dev.index, _stream(dev))
    _lib.check(st, "lcn_xcorrvol_argmax")
    return (y, std, idx, best, vol) if return_volume else (y, std, idx, best)


def lcn_normalize(img, kernel_size=4, epsilon=0.01):
    """Additive: the data generator's LCN, `lcn.normalize(img, kernel_size, epsilon)` of data/lcn/lcn.pyx:16-58
    (two-pass mean / std, zero border of width kernel_size, returns (lcn, raw std)); img [H,W] or [N,H,W]."""
    _check(img, "img", (torch.float32,))
    squeeze = img.dim() == 2
    a = img.unsqueeze(0) if squeeze else img
    if a.dim() != 3:
        raise RuntimeError("lcn_normalize expects [H,W] or [N,H,W]")
    N, H, W = a.shape
    dev = a.device
    out, std = torch.empty_like(a), torch.empty_like(a)
    st = _lib.lib().ctd_lcn_datagen_f32(_ptr(a), _ptr(out), _ptr(std), N, H, W, int(kernel_size), float(epsilon), dev.index,
                                        _stream(dev))
    _lib.check(st, "lcn_normalize")
    return (out[0], std[0]) if squeeze else (out, std)


# --------------------------------------------------------------------------------------
# Photometric block loss (reference: PhotometricLossFunction, functions.py:79-118)
# --------------------------------------------------------------------------------------
def _photo_args(es, ta):
    _check(es, "es")
    _check(ta, "ta")
    if es.dim() != 4 or es.shape != ta.shape or es.dtype != ta.dtype:
        raise RuntimeError("photometric_loss expects es and ta of the same [B,C,H,W] shape and dtype")
    return _same_device(es, ta)


def _photo_fast(es, block_size, algo):
    """algo 'fast' (default) = tolerance-level kernels (f32, odd block sizes up to 9); anything else they do not
    cover runs the reference-order kernels."""
    algo = algo or os.environ.get("CTD_PHOTO_ALGO", "fast")
    if algo not in _ALGOS:
        raise RuntimeError("unknown algo %r" % (algo,))
    return algo == "fast" and es.dtype == torch.float32 and int(block_size) in (3, 5, 7, 9)


class PhotometricLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, es, ta, block_size, type, eps, algo=None):
        dev = _photo_args(es, ta)
        ctx.save_for_backward(es, ta)
        ctx.block_size = block_size
        ctx.type = type
        ctx.eps = eps
        ctx.fast = _photo_fast(es, block_size, algo)
        B, C, H, W = es.shape
        out = torch.empty((B, 1, H, W), dtype=es.dtype, device=dev)
        L = _lib.lib()
        fn = L.ctd_photometric_fwd_f32 if es.dtype == torch.float32 else L.ctd_photometric_fwd_f64
        if ctx.fast:
            fn = L.ctd_photometric_fwd_fast_f32
        st = fn(_ptr(es), _ptr(ta), _ptr(out), B, C, H, W, int(block_size), int(type), float(eps), dev.index,
                _stream(dev))
        _lib.check(st, "photometric_loss_forward")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        es, ta = ctx.saved_tensors
        grad_out = grad_out.contiguous()                      # functions.py:99
        _check(grad_out, "grad_out")
        dev = es.device
        B, C, H, W = es.shape
        # the gather backward writes every element: no zero-fill needed (the reference scatters into at::zeros)
        grad_es = torch.empty_like(es)
        L = _lib.lib()
        fn = L.ctd_photometric_bwd_f32 if es.dtype == torch.float32 else L.ctd_photometric_bwd_f64
        if ctx.fast:
            fn = L.ctd_photometric_bwd_fast_f32
        st = fn(_ptr(es), _ptr(ta), _ptr(grad_out), _ptr(grad_es), B, C, H, W, int(ctx.block_size), int(ctx.type),
                float(ctx.eps), dev.index, _stream(dev))
        _lib.check(st, "photometric_loss_backward")
        return grad_es, None, None, None, None, None


_PHOTO_TYPES = {"mse": 0, "sad": 1, "census_mse": 2, "census_sad": 3}


def photometric_loss(es, ta, block_size, type='mse', eps=0.1, algo=None):
    """[B,C,H,W] x2 -> [B,1,H,W]: mean over a block_size^2 replicate-clamped block, summed over channels, of
    (es-ta)^2 | |es-ta| | soft-census squared / absolute difference.  Gradient flows to `es` only.
    algo (additive): 'fast' (default, or env CTD_PHOTO_ALGO) = tolerance-level kernels, ~10x faster for the census
    types; 'exact' = reference operation order, bit-identical to the reference CPU build."""
    type = type.lower()
    if type not in _PHOTO_TYPES:
        raise Exception('invalid loss type')                  # functions.py:117
    return PhotometricLossFunction.apply(es, ta, block_size, _PHOTO_TYPES[type], eps, algo)


class PatternLossFunction(torch.autograd.Function):
    """Additive: the whole of RectifiedPatternSimilarityLoss.tforward (networks.py:358-378) as one
    forward and one backward kernel -- warp, block loss (block 9), masked mean.  Tolerance level (fast kernels).
    Returns (val, pattern_proj, terms) with terms = [sum(mask*diff), sum(mask), val] for cross-rank reduction."""

    @staticmethod
    def forward(ctx, disp, im, mask, pattern, type, eps):
        for t, name in ((disp, "disp"), (im, "im"), (pattern, "pattern")) + (((mask, "mask"),) if mask is not None else ()):
            _check(t, name, (torch.float32,))
        dev = _same_device(disp, im, pattern, *(() if mask is None else (mask,)))
        if disp.dim() != 4 or disp.shape[1] != 1 or im.shape != disp.shape or (mask is not None and mask.shape != disp.shape):
            raise RuntimeError("pattern_loss expects disp, im (and mask) of the same [B,1,H,W] shape")
        B, _, H, W = disp.shape
        if pattern.numel() != H * W:
            raise RuntimeError("pattern_loss expects a single-channel pattern with H*W elements")
        L = _lib.lib()
        proj = torch.empty_like(disp)
        terms = torch.empty(3, dtype=torch.float32, device=dev)
        ws = _workspace(L.ctd_pattern_loss_workspace_bytes(B, H, W), dev)
        st = L.ctd_pattern_loss_fwd_f32(_ptr(disp), _ptr(im), _ptr(mask), _ptr(pattern), _ptr(proj), _ptr(terms), B, H, W,
                                        int(type), float(eps), _ptr(ws), ws.numel(), dev.index, _stream(dev))
        _lib.check(st, "pattern_loss_forward")
        ctx.save_for_backward(disp, im, pattern, terms, *(() if mask is None else (mask,)))
        ctx.has_mask = mask is not None
        ctx.type, ctx.eps = int(type), float(eps)
        return terms[2].clone(), proj, terms

    @staticmethod
    def backward(ctx, grad_val, grad_proj, grad_terms):
        saved = ctx.saved_tensors
        disp, im, pattern, terms = saved[:4]
        mask = saved[4] if ctx.has_mask else None
        dev = disp.device
        B, _, H, W = disp.shape
        # the kernel applies go[p] = gv * mask[p] / den; gradients arriving at the numerator (cross-rank ratio of
        # sums, sharding.reduce_ratio) or at terms[2] fold into the same scalar.  den does not depend on disp.
        gv = torch.zeros(1, dtype=torch.float32, device=dev)
        if grad_val is not None:
            gv = gv + grad_val.reshape(1)
        if grad_terms is not None:
            gv = gv + grad_terms[2:3] + grad_terms[0:1] * terms[1:2]
        gp = grad_proj.contiguous() if grad_proj is not None else None
        grad_disp = torch.empty_like(disp)
        st = _lib.lib().ctd_pattern_loss_bwd_f32(_ptr(disp), _ptr(im), _ptr(mask), _ptr(pattern), _ptr(terms), _ptr(gv),
                                                 _ptr(gp), _ptr(grad_disp), B, H, W, ctx.type, ctx.eps, dev.index,
                                                 _stream(dev))
        _lib.check(st, "pattern_loss_backward")
        return grad_disp, None, None, None, None, None


def pattern_loss(disp, im, mask, pattern, type='census_sad', eps=0.5):
    """Fused pattern similarity loss: (val, pattern_proj, terms); gradient flows to `disp` only."""
    type = type.lower()
    if type not in _PHOTO_TYPES:
        raise Exception('invalid loss type')
    return PatternLossFunction.apply(disp, im, mask, pattern, _PHOTO_TYPES[type], eps)


class PatternLossMultiFunction(torch.autograd.Function):
    """Additive: the pattern similarity loss of all pyramid levels in one forward and one backward
    launch.  apply(type, eps, n, disp_0..disp_{n-1}, im_0.., mask_0.. (None allowed), pattern_0..) ->
    (vals [n], terms [n,3], proj_0..proj_{n-1})."""

    @staticmethod
    def _levels(n, disps, ims, masks, patterns, projs=None, grad_projs=None, grad_disps=None):
        arr = (_lib.PatternLevel * n)()
        for l in range(n):
            B, _, H, W = disps[l].shape
            arr[l] = _lib.PatternLevel(_ptr(disps[l]), _ptr(ims[l]), _ptr(masks[l]), _ptr(patterns[l]),
                                       _ptr(projs[l]) if projs else None, _ptr(grad_projs[l]) if grad_projs else None,
                                       _ptr(grad_disps[l]) if grad_disps else None, B, H, W)
        return arr

    @staticmethod
    def forward(ctx, type, eps, n, *tensors):
        disps, ims, masks, patterns = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n], tensors[3 * n:4 * n]
        for l in range(n):
            for t, name in ((disps[l], "disp"), (ims[l], "im"), (patterns[l], "pattern")) + (
                    ((masks[l], "mask"),) if masks[l] is not None else ()):
                _check(t, "%s[%d]" % (name, l), (torch.float32,))
            if disps[l].dim() != 4 or disps[l].shape[1] != 1 or ims[l].shape != disps[l].shape or (
                    masks[l] is not None and masks[l].shape != disps[l].shape):
                raise RuntimeError("pattern_loss_multi: level %d expects disp, im (and mask) of one [B,1,H,W] shape" % l)
            if patterns[l].numel() != disps[l].shape[2] * disps[l].shape[3]:
                raise RuntimeError("pattern_loss_multi: level %d pattern must have H*W elements" % l)
        dev = _same_device(*disps, *ims, *patterns, *[m for m in masks if m is not None])
        L = _lib.lib()
        projs = [torch.empty_like(d) for d in disps]
        terms = torch.empty((n, 3), dtype=torch.float32, device=dev)
        levels = PatternLossMultiFunction._levels(n, disps, ims, masks, patterns, projs)
        ws = _workspace(L.ctd_pattern_loss_multi_workspace_bytes(n, levels), dev)
        st = L.ctd_pattern_loss_multi_fwd_f32(n, levels, _ptr(terms), int(type), float(eps), _ptr(ws), ws.numel(), dev.index,
                                              _stream(dev))
        _lib.check(st, "pattern_loss_multi_forward")
        ctx.save_for_backward(terms, *disps, *ims, *patterns, *[m for m in masks if m is not None])
        ctx.n, ctx.type, ctx.eps = n, int(type), float(eps)
        ctx.has_mask = [m is not None for m in masks]
        return (terms[:, 2].clone(), terms) + tuple(projs)

    @staticmethod
    def backward(ctx, grad_vals, grad_terms, *grad_projs):
        n = ctx.n
        saved = ctx.saved_tensors
        terms = saved[0]
        disps, ims, patterns = saved[1:1 + n], saved[1 + n:1 + 2 * n], saved[1 + 2 * n:1 + 3 * n]
        rest = list(saved[1 + 3 * n:])
        masks = [rest.pop(0) if h else None for h in ctx.has_mask]
        dev = terms.device
        # gradients arriving at the numerators / ratios fold into one scalar per level (see PatternLossFunction)
        gv = torch.zeros(n, dtype=torch.float32, device=dev)
        if grad_vals is not None:
            gv = gv + grad_vals
        if grad_terms is not None:
            gv = gv + grad_terms[:, 2] + grad_terms[:, 0] * terms[:, 1]
        gv = gv.contiguous()
        gps = [g.contiguous() if g is not None else None for g in grad_projs]
        grad_disps = [torch.empty_like(d) for d in disps]
        levels = PatternLossMultiFunction._levels(n, disps, ims, masks, patterns, None, gps, grad_disps)
        st = _lib.lib().ctd_pattern_loss_multi_bwd_f32(n, levels, _ptr(terms), _ptr(gv), ctx.type, ctx.eps, dev.index,
                                                       _stream(dev))
        _lib.check(st, "pattern_loss_multi_backward")
        return (None, None, None) + tuple(grad_disps) + (None,) * (3 * n)


def pattern_loss_multi(disps, ims, masks, patterns, type='census_sad', eps=0.5):
    """Fused pattern similarity loss of several pyramid levels (lists of per-level tensors; masks may hold None)
    in one launch each way: (vals [n], terms [n,3], [pattern_proj per level])."""
    type = type.lower()
    if type not in _PHOTO_TYPES:
        raise Exception('invalid loss type')
    n = len(disps)
    if not (len(ims) == len(masks) == len(patterns) == n) or n < 1 or n > 8:
        raise RuntimeError("pattern_loss_multi expects 1..8 levels with one disp, im, mask and pattern each")
    out = PatternLossMultiFunction.apply(_PHOTO_TYPES[type], eps, n, *[d.contiguous() for d in disps],
                                         *[i.contiguous() for i in ims],
                                         *[None if m is None else m.contiguous() for m in masks], *patterns)
    return out[0], out[1], list(out[2:])


def photometric_loss_pytorch(es, ta, block_size, type='mse', eps=0.1):
    """Stock-PyTorch formulation of the same loss (replicate pad + unfold), kept as the independent
    second opinion the reference ships next to its kernels (functions.py:120-147)."""
    type = type.lower()
    if type not in _PHOTO_TYPES:
        raise Exception('invalid loss type')
    p = block_size // 2
    B, C, H, W = es.shape

    def windows(x):
        xp = torch.nn.functional.pad(x, (p, p, p, p), mode='replicate')
        return torch.nn.functional.unfold(xp, kernel_size=block_size).view(B, C, -1, H, W)

    ew, tw = windows(es), windows(ta)
    if type in ('mse', 'sad'):
        diff = ew - tw
    else:
        def soft(d):
            return 0.5 * (1 + d / torch.sqrt(d * d + eps))
        diff = soft(ew - es.unsqueeze(2)) - soft(tw - ta.unsqueeze(2))
    term = diff * diff if type.endswith('mse') else diff.abs()
    return term.reshape(B, -1, H, W).sum(dim=1, keepdim=True) / block_size ** 2


def costvol(im, pattern, n_disps, block_size, type='sad', eps=0.1, algo=None):
    """Additive: SAD / MSE / soft-census block cost volume between frames and the pattern shifted by d
    (SURVEY 8a/A6): cost[f,d] = photometric_loss(P_d, im[f]) with P_d[h,x] = P[h, clamp(x-d)].
    im [N,H,W] | [H,W] f32, pattern [H,W] | [N,H,W] -> [N,D,H,W] | [D,H,W]; argmin over d is the best match."""
    _check(im, "im", (torch.float32,))
    _check(pattern, "pattern", (torch.float32,))
    type = type.lower()
    if type not in _PHOTO_TYPES:
        raise Exception('invalid loss type')
    squeeze = im.dim() == 2
    a = im.unsqueeze(0) if squeeze else im
    if a.dim() != 3 or pattern.dim() not in (2, 3) or tuple(pattern.shape[-2:]) != tuple(a.shape[-2:]):
        raise RuntimeError("costvol expects im [N,H,W] or [H,W] and pattern [H,W] or [N,H,W]")
    dev = _same_device(a, pattern)
    N, H, W = a.shape
    stride = 0 if pattern.dim() == 2 else H * W
    D = int(n_disps)
    out = torch.empty((N, D, H, W), dtype=torch.float32, device=dev)
    L = _lib.lib()
    if _photo_fast(a, block_size, algo):
        # SAD / MSE, block 9: the separable path stages padded operand planes in a workspace (0 bytes: other kernels)
        nws = L.ctd_costvol_workspace_bytes(N, H, W, D, int(block_size), _PHOTO_TYPES[type], 1 if stride else 0)
        ws = _workspace(nws, dev) if nws else None
        st = L.ctd_costvol_fast_f32(_ptr(a), _ptr(pattern), stride, _ptr(out), N, H, W, D, int(block_size),
                                    _PHOTO_TYPES[type], float(eps), _ptr(ws), nws, dev.index, _stream(dev))
    else:
        st = L.ctd_costvol_f32(_ptr(a), _ptr(pattern), stride, _ptr(out), N, H, W, D, int(block_size),
                               _PHOTO_TYPES[type], float(eps), dev.index, _stream(dev))
    _lib.check(st, "costvol")
    return out[0] if squeeze else out


# --------------------------------------------------------------------------------------
# Fused loss kernels (reference: stock-PyTorch modules of model/networks.py; additive API)
# --------------------------------------------------------------------------------------
def _f32(t, name):
    _check(t, name, (torch.float32,))
    return t


class DispToDepthFunction(torch.autograd.Function):
    """depth = baseline_focal / (relu(disp) + 1e-12)   (networks.DispToDepth.tforward, networks.py:318-321)"""

    @staticmethod
    def forward(ctx, disp, baseline_focal):
        disp = _f32(disp.contiguous(), "disp")
        ctx.save_for_backward(disp)
        ctx.bf = float(baseline_focal)
        depth = torch.empty_like(disp)
        dev = disp.device
        st = _lib.lib().ctd_disp_to_depth_fwd_f32(_ptr(disp), _ptr(depth), disp.numel(), ctx.bf, dev.index, _stream(dev))
        _lib.check(st, "disp_to_depth")
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        (disp,) = ctx.saved_tensors
        grad_depth = _f32(grad_depth.contiguous(), "grad_depth")
        grad = torch.empty_like(disp)
        dev = disp.device
        st = _lib.lib().ctd_disp_to_depth_bwd_f32(_ptr(disp), _ptr(grad_depth), _ptr(grad), disp.numel(), ctx.bf,
                                                  dev.index, _stream(dev))
        _lib.check(st, "disp_to_depth backward")
        return grad, None


def disp_to_depth(disp, baseline_focal):
    return DispToDepthFunction.apply(disp, baseline_focal)


def idx_to_depth(idx, baseline_focal, disp_offset=0.0):
    """Additive: depth of the int64 disparity indices `xcorrvol_argmax` returns, `baseline_focal / (relu(idx +
    disp_offset) + 1e-12)`, same shape as `idx` (f32).  Not differentiable."""
    _check(idx, "idx", (torch.int64,))
    depth = torch.empty(idx.shape, dtype=torch.float32, device=idx.device)
    dev = idx.device
    st = _lib.lib().ctd_idx_to_depth_f32(_ptr(idx), _ptr(depth), idx.numel(), float(baseline_focal), float(disp_offset),
                                         dev.index, _stream(dev))
    _lib.check(st, "idx_to_depth")
    return depth


class DisparityLossFunction(torch.autograd.Function):
    """Sobel 5x5 + edge-aware disparity loss (networks.DisparityLoss.tforward, networks.py:395-412) -> scalar."""

    @staticmethod
    def forward(ctx, disp, edge):
        disp = _f32(disp.contiguous(), "disp")
        if disp.dim() != 4 or disp.shape[1] != 1:
            raise RuntimeError("disparity_loss expects disp [B,1,H,W]")
        if edge is not None:
            edge = _f32(edge.contiguous(), "edge")
            if edge.shape != disp.shape:
                raise RuntimeError("edge must have the shape of disp")
            _same_device(disp, edge)
        B, _, H, W = disp.shape
        dev = disp.device
        L = _lib.lib()
        ws = _workspace(L.ctd_disparity_loss_workspace_bytes(B, H, W), dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        st = L.ctd_disparity_loss_fwd_f32(_ptr(disp), _ptr(edge), _ptr(loss), B, H, W, _ptr(ws), ws.numel(), dev.index,
                                          _stream(dev))
        _lib.check(st, "disparity_loss")
        ctx.save_for_backward(disp, edge)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        disp, edge = ctx.saved_tensors
        B, _, H, W = disp.shape
        dev = disp.device
        L = _lib.lib()
        ws = _workspace(L.ctd_disparity_loss_workspace_bytes(B, H, W), dev)
        gl = grad_loss.to(torch.float32).contiguous()
        grad_disp = torch.empty_like(disp)
        want_edge = edge is not None and ctx.needs_input_grad[1]
        grad_edge = torch.empty_like(edge) if want_edge else None
        st = L.ctd_disparity_loss_bwd_f32(_ptr(disp), _ptr(edge), _ptr(gl), _ptr(grad_disp), _ptr(grad_edge), B, H, W,
                                          _ptr(ws), ws.numel(), dev.index, _stream(dev))
        _lib.check(st, "disparity_loss backward")
        return grad_disp, grad_edge


def disparity_loss(disp, edge=None):
    return DisparityLossFunction.apply(disp, edge)


class GeometricLossFunction(torch.autograd.Function):
    """Symmetric two-view geometric loss (networks.ProjectionDepthSimilarityLoss.tforward, networks.py:500-503):
    fwd(depth0 -> view 1) + fwd(depth1 -> view 0), each a mean of clamped |projected depth - sampled depth|."""

    @staticmethod
    def forward(ctx, depth0, depth1, ray, K, R0, t0, R1, t1, clamp):
        ts = [_f32(x.contiguous(), n) for x, n in ((depth0, "depth0"), (depth1, "depth1"), (ray, "ray"), (K, "K"),
                                                   (R0, "R0"), (t0, "t0"), (R1, "R1"), (t1, "t1"))]
        depth0, depth1, ray, K, R0, t0, R1, t1 = ts
        dev = _same_device(*ts)
        if depth0.dim() != 4 or depth0.shape[1] != 1 or depth0.shape != depth1.shape:
            raise RuntimeError("geometric_loss expects depth0, depth1 [B,1,H,W]")
        B, _, H, W = depth0.shape
        if ray.numel() != H * W * 3 or K.numel() != 9 or R0.numel() != B * 9 or R1.numel() != B * 9 or \
                t0.numel() != B * 3 or t1.numel() != B * 3:
            raise RuntimeError("geometric_loss: ray [H*W,3], K [3,3], R [B,3,3], t [B,3] expected")
        L = _lib.lib()
        loss = torch.empty((), dtype=torch.float32, device=dev)
        c = float(clamp)
        if L.ctd_geometric_workspace_bytes(2 * B, H, W) > 0:
            # both directions in one launch, the means formed by its last workgroup (tickets: zeroed words per stream)
            ws = _workspace(L.ctd_geometric_workspace_bytes(2 * B, H, W), dev)
            _call_with_ticket(lambda tk: L.ctd_geometric_sym_fwd_f32(
                _ptr(depth0), _ptr(depth1), _ptr(ray), _ptr(K), _ptr(R0), _ptr(t0), _ptr(R1), _ptr(t1), _ptr(loss), B, H, W,
                c, _ptr(ws), ws.numel(), _ptr(tk), dev.index, _stream(dev)), _ticket(dev), "geometric_loss")
        else:
            ws = _workspace(L.ctd_geometric_workspace_bytes(B, H, W), dev)
            st = L.ctd_geometric_fwd_f32(_ptr(depth0), _ptr(depth1), _ptr(ray), _ptr(K), _ptr(R0), _ptr(t0), _ptr(R1), _ptr(t1),
                                         _ptr(loss), 0, B, H, W, c, _ptr(ws), ws.numel(), dev.index, _stream(dev))
            _lib.check(st, "geometric_loss")
            st = L.ctd_geometric_fwd_f32(_ptr(depth1), _ptr(depth0), _ptr(ray), _ptr(K), _ptr(R1), _ptr(t1), _ptr(R0), _ptr(t0),
                                         _ptr(loss), 1, B, H, W, c, _ptr(ws), ws.numel(), dev.index, _stream(dev))
            _lib.check(st, "geometric_loss")
        ctx.save_for_backward(depth0, depth1, ray, K, R0, t0, R1, t1)
        ctx.clamp = c
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        depth0, depth1, ray, K, R0, t0, R1, t1 = ctx.saved_tensors
        B, _, H, W = depth0.shape
        dev = depth0.device
        L = _lib.lib()
        gl = grad_loss.to(torch.float32).contiguous()
        g0 = torch.zeros_like(depth0)          # both receive bilinear scatter (atomics) from the other direction
        g1 = torch.zeros_like(depth1)
        st = L.ctd_geometric_bwd_f32(_ptr(depth0), _ptr(depth1), _ptr(ray), _ptr(K), _ptr(R0), _ptr(t0), _ptr(R1), _ptr(t1),
                                     _ptr(gl), _ptr(g0), 1, _ptr(g1), B, H, W, ctx.clamp, dev.index, _stream(dev))
        _lib.check(st, "geometric_loss backward")
        st = L.ctd_geometric_bwd_f32(_ptr(depth1), _ptr(depth0), _ptr(ray), _ptr(K), _ptr(R1), _ptr(t1), _ptr(R0), _ptr(t0),
                                     _ptr(gl), _ptr(g1), 1, _ptr(g0), B, H, W, ctx.clamp, dev.index, _stream(dev))
        _lib.check(st, "geometric_loss backward")
        return g0, g1, None, None, None, None, None, None, None


def geometric_loss(depth0, depth1, ray, K, R0, t0, R1, t1, clamp=-1):
    return GeometricLossFunction.apply(depth0, depth1, ray, K, R0, t0, R1, t1, clamp)
