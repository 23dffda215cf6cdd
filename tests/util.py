import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    """Float parity rule of SURVEY 8d: |a-b| <= rtol*|b| + atol."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b) - (rtol * np.abs(b) + atol)
    worst = err.max() if err.size else 0.0
    assert worst <= 0, "%s: %d/%d outside tol, max |a-b|=%.3e" % (what, (err > 0).sum(), err.size, np.abs(a - b).max())
