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
