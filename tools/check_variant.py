"""Bit-equality of a library variant against the in-tree library on the ranked call (volume + indices + best) and the
volume-free call: python tools/check_variant.py variant.so"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads
from connecting_the_dots_amd import _lib

def outputs(path, N, H, W, D):
    _lib._lib = None
    if not hasattr(_lib, "_IN_TREE"):
        _lib._IN_TREE = _lib.LIB_PATH
    _lib.LIB_PATH = os.path.abspath(path) if path else _lib._IN_TREE
    from connecting_the_dots_amd import torchext as te
    fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
    pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
    x, _ = te.lcn(fr, 5, 0.05)
    p = te.lcn(pat, 5, 0.05)[0][0].contiguous()
    idx, best, vol = te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast")
    idx2, best2 = te.xcorrvol_argmax(x, p, D, 9, algo="fast")
    plain = te.xcorrvol_batch(x, p, D, 9, algo="fast")
    torch.cuda.synchronize()
    return [t.clone() for t in (idx, best, vol, idx2, best2, plain)]

for shape in ((16, 432, 512, 128), (2, 100, 1024, 256), (3, 45, 300 // 4 * 4, 27), (1, 31, 256, 14)):
    a = outputs("", *shape)
    b = outputs(sys.argv[1], *shape)
    ok = [bool(torch.equal(u, v)) for u, v in zip(a, b)]
    print(shape, "equal:", ok, flush=True)
    assert all(ok)
print("variant == in-tree")
