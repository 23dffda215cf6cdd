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


def render_scene(seed, H=48, W=64, n_boxes=4, wall=True):
    """Small structured-light scene for the renderer tests: a slanted back wall plus a few random boxes in front
    of it, a pinhole camera at the origin and a projector 7.5 cm to its side (the reference's baseline,
    data/create_syn_data.py:228-230), a random RGB projector pattern.  Returns the arguments of render_mesh_proj."""
    rs = np.random.RandomState(seed)
    verts, faces = [], []

    def quad(p0, p1, p2, p3):
        b = len(verts)
        verts.extend([p0, p1, p2, p3])
        faces.extend([[b, b + 1, b + 2], [b, b + 2, b + 3]])

    if wall:
        quad([-3, -2, 3.0], [3, -2, 3.5], [3, 2, 3.5], [-3, 2, 3.0])              # back wall (else: rays that miss)
    for _ in range(n_boxes):
        c = np.array([rs.uniform(-0.8, 0.8), rs.uniform(-0.5, 0.5), rs.uniform(1.2, 2.4)])
        s = rs.uniform(0.1, 0.3, size=3)
        corners = [c + s * np.array([dx, dy, dz]) for dx in (-1, 1) for dy in (-1, 1) for dz in (-1, 1)]
        for a, b_, c_, d in ((0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)):
            quad(corners[a], corners[b_], corners[c_], corners[d])
    verts = np.array(verts, np.float32)
    faces = np.array(faces, np.int32)
    colors = rs.uniform(0.3, 1.0, size=verts.shape).astype(np.float32)
    f = 0.9 * W
    K = np.array([[f, 0, W / 2 - 0.5], [0, f, H / 2 - 0.5], [0, 0, 1]], np.float32)
    R = np.eye(3, dtype=np.float32)
    cam = (K, R, np.zeros(3, np.float32), W, H)
    proj = (K, R, np.array([0.075, 0, 0], np.float32), W, H)          # x_proj = x_cam + baseline
    pattern = rs.uniform(0, 1, size=(H, W, 3)).astype(np.float32)
    shader = (0.5, 1.5, 0.0, 10.0)                                    # create_syn_data.py:155
    return dict(verts=verts, colors=colors, faces=faces, cam=cam, proj=proj, shader=shader, pattern=pattern,
                d_alpha=0.0, d_beta=0.35)


def render_normals(sc, seed):
    """per-vertex normals for the plain mesh renderer: the area-weighted face normals gathered per vertex plus a
    seeded jitter, deliberately left unnormalised (the reference interpolates them as given)"""
    v, f = sc["verts"].astype(np.float64), sc["faces"]
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-9)
    n += np.random.RandomState(seed).uniform(-0.2, 0.2, size=n.shape)
    return n.astype(np.float32)


def small_pose(rs):
    """a small seeded rigid motion (rotation by ~0.01 rad about a random axis, 2 cm translation)"""
    ax = rs.randn(3) * 0.01
    th = np.linalg.norm(ax)
    k = ax / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    return R.astype(np.float32), (rs.randn(3) * 0.02).astype(np.float32)


def track_batch(seed, tl, B, H, W, D, n_scales=4, block=(48, 64)):
    """A synthetic training batch shaped like the reference's TrackSynDataset samples after Worker.copy_data
    (model/exp_synphge.py:93-115): per scale s the raw IR frames im{s} [tl,B,1,H_s,W_s] (scale s = 2x2 box-averaged
    scale s-1, the way data/dataset.py builds its pyramids), the ground-truth disparity's gradient magnitude grad{s}
    (scales 0..2), sample ids, and per-frame poses R [tl,B,3,3], t [tl,B,3].  numpy arrays."""
    rs = np.random.RandomState(seed)
    pat = syn_dot_pattern(H, W, seed=42)
    ims = np.zeros((tl, B, 1, H, W), np.float32)
    disps = np.zeros((tl, B, 1, H, W), np.float32)
    R = np.zeros((tl, B, 3, 3), np.float32)
    t = np.zeros((tl, B, 3), np.float32)
    for i in range(tl):
        for b in range(B):
            ims[i, b, 0], d = synth_ir(pat, rs, D, block=block)
            disps[i, b, 0] = d
            R[i, b], t[i, b] = small_pose(rs)
    out = {"id": np.arange(B, dtype=np.int64), "R": R, "t": t, "pattern": pat, "disp0": disps}
    im, dd = ims, disps
    for s in range(n_scales):
        out["im%d" % s] = im
        if s < 3:
            gy, gx = np.gradient(dd, axis=(3, 4))
            out["grad%d" % s] = np.sqrt(gx * gx + gy * gy).astype(np.float32)
        im = im.reshape(tl, B, 1, im.shape[3] // 2, 2, im.shape[4] // 2, 2).mean(axis=(4, 6))
        dd = dd.reshape(tl, B, 1, dd.shape[3] // 2, 2, dd.shape[4] // 2, 2).mean(axis=(4, 6)) / 2
    return out
