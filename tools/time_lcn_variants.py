"""LCN algo='fast' per library variant (tile shapes: tools/build_variant.sh lcn_WxH csrc/lcn.hip -DCTD_LCN_FAST_TW=W -DCTD_LCN_FAST_TH=H):
    python tools/time_lcn_variants.py [lib.so ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(path, shape):
    from connecting_the_dots_amd import _lib
    _lib._lib = None
    if not hasattr(_lib, "_IN_TREE"):
        _lib._IN_TREE = _lib.LIB_PATH
    _lib.LIB_PATH = os.path.abspath(path) if path else _lib._IN_TREE
    from connecting_the_dots_amd import torchext as te
    torch.manual_seed(0)
    x = torch.rand(*shape, device="cuda")
    ref, _ = te.lcn(x, 5, 0.05, algo="exact")
    y, _ = te.lcn(x, 5, 0.05, algo="fast")
    ok = bool(((y - ref).abs() <= ref.abs() * 1e-5 + 1e-6).all())
    for _ in range(300): te.lcn(x, 5, 0.05, algo="fast")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): te.lcn(x, 5, 0.05, algo="fast")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e6, ok

for rep in range(2):
    for p in (sys.argv[1:] or [""]):
        a, ok1 = run(p, (16, 1, 432, 512))
        b, ok2 = run(p, (1, 1, 1024, 1024))
        print("%-24s 16x432x512 %.2f us   1x1024x1024 %.2f us   within tolerance: %s" % (os.path.basename(p or "in-tree 32x32"), a, b, ok1 and ok2), flush=True)
