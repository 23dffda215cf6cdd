"""Times LCN and the pre-pass-carrying volume call of experimental builds (tools/variants/*.so) and checks their outputs
against the in-tree library bit for bit:   python tools/time_side.py [lib.so ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads

H, W, D, N = 432, 512, 128, 16
ref = {}


def run(path):
    from connecting_the_dots_amd import _lib
    _lib._lib = None
    if not hasattr(_lib, "_IN_TREE"):
        _lib._IN_TREE = _lib.LIB_PATH
    _lib.LIB_PATH = path or _lib._IN_TREE
    from connecting_the_dots_amd import torchext as te
    fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
    pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
    p = te.lcn(pat, 5, 0.05)[0][0].contiguous()
    out = {}

    def timed(name, fn, n_warm=300, n=100):
        for _ in range(n_warm):
            r = fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t) / n * 1e3
        return r

    y, sd = timed("lcn", lambda: te.lcn(fr, 5, 0.05))
    idx, best = timed("argmax(volume-free)", lambda: te.xcorrvol_argmax(y, p, D, 9), 100, 40)
    res = dict(y=y, sd=sd, idx=idx, best=best)
    same = ""
    if not path:
        ref.update(res)
    elif ref:
        same = "  bit-equal to in-tree: " + ", ".join("%s=%s" % (k, bool(torch.equal(v, ref[k]))) for k, v in res.items())
    print("%-26s " % os.path.basename(path or "in-tree") + "  ".join("%s %.4f ms" % kv for kv in out.items()) + same,
          flush=True)


for _ in range(2):
    for p in [""] + sys.argv[1:]:
        run(p)
