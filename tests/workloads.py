"""Seeded synthetic workloads shared by tests/, tests/golden/make_golden.py and bench.py.

Everything here is regenerated from numpy's frozen legacy RandomState streams, so the
committed golden outputs can be checked on any box without shipping the inputs.
"""
import numpy as np


def uniform_frame(seed, H, W, C=1):
    """RandomState(seed) uniform [0,1) frame, f32 [C,H,W]."""
    return np.random.RandomState(seed).rand(C, H, W).astype(np.float32)


def syn_dot_pattern(H, W, seed=42):
    """The reference's synthetic pattern rule, data/commons.py:8-11 (uniform < 0.1)."""
    rs = np.random.RandomState(seed)
    return (rs.uniform(0, 1, size=(H, W)) < 0.1).astype(np.float32)


def synth_ir(pattern01, rs, D=128, block=(48, 64)):
    """Config-1 style IR frame: the pattern seen under a piecewise-constant
    disparity map, blended 0.6*pattern + 0.4*ambient + N(0,(3/255)^2), clipped to [0,1].
    Convention of xcorrvol (ext.h:152): pixel w of the frame matches pattern column w - d.
    Returns (ir f32 [H,W], disparity int64 [H,W])."""
    H, W = pattern01.shape
    disp = np.zeros((H, W), np.int64)
    for by in range(0, H, block[0]):
        for bx in range(0, W, block[1]):
            disp[by:by + block[0], bx:bx + block[1]] = rs.randint(0, D)
    cols = np.clip(np.arange(W)[None, :] - disp, 0, W - 1)
    shifted = np.take_along_axis(pattern01, cols, axis=1)
    ambient = rs.uniform(0, 1, size=(H, W))
    ir = 0.6 * shifted + 0.4 * ambient + rs.normal(0, 3.0 / 255, size=(H, W))
    return np.clip(ir, 0, 1).astype(np.float32), disp


def nn_case(seed, dt, n0=300, n1=257):
    """seeded point clouds with exact duplicates (ties) for nn / crosscheck; shared with the tests"""
    rs = np.random.RandomState(seed)
    a = rs.normal(size=(n0, 3)).astype(dt)
    b = rs.normal(size=(n1, 3)).astype(dt)
    b[10] = b[200]
    a[5] = b[200]
    a[7] = 1e6            # farther than sqrt(1e9) from everything: no match, -1
    return a, b


def proj_case(seed, dt, B=2, H=24, W=32):
    rs = np.random.RandomState(seed)
    K = np.array([[30., 0, 16], [0, 30, 12], [0, 0, 1]], dt)
    z = rs.uniform(1, 3, size=(B, H, W)).astype(dt)
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    xyz1 = np.stack([(u - 16) / 30 * z, (v - 12) / 30 * z, z], -1).astype(dt)
    xyz0 = (xyz1 + rs.normal(scale=0.05, size=xyz1.shape)).astype(dt)
    xyz0[0, 0, 0] = [0, 0, 0]      # d = 0: NaN projection
    xyz0[0, 0, 1] = [1, 1, 0]      # infinite projection
    xyz0[1, 3, 4] = [-50, 0, 1]    # projects far left of the image
    return xyz0, xyz1, K
