"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py

Sources of truth:
  * native ops  : /root/reference/torchext/ext/ext_cpu.cpp compiled unmodified by
                  oracle/build_ref.py (g++ -O3, no FMA contraction)
  * Python ops  : /root/reference/model/networks.py imported via oracle/ref_python.py
  * Cython LCN  : /root/reference/data/lcn/lcn.pyx cythonized under /tmp (if Cython works)
  * pattern     : /root/reference/data/kinect_pattern.png, centre-cropped with the
                  rule of data/commons.py:21-24 (a data file, stored as uint8)

Only inputs and expected outputs are written; no reference source is stored.
The files are small (<1 MB each, compressed) and are committed.
"""
import hashlib
import os
import subprocess
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402

from oracle import build_ref, ref_python  # noqa: E402
from tests import workloads  # noqa: E402

warnings.filterwarnings("ignore")
torch.set_num_threads(4)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    p = os.path.join(OUT, name + ".npz")
    np.savez_compressed(p, **arrs)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(p) / 1024))


def gen_xcorrvol(ext):
    out = {}
    cases = []
    rs = np.random.RandomState(7)
    k = 0
    for dt in ("float32", "float64"):
        for (C, H, W, D, bs, kind) in [(1, 16, 40, 8, 9, "uniform"), (2, 24, 33, 16, 5, "normal"),
                                       (1, 12, 20, 24, 4, "uniform"), (1, 20, 70, 40, 9, "normal"),
                                       (3, 9, 11, 16, 3, "uniform")]:
            if kind == "uniform":
                a, b = rs.rand(C, H, W), rs.rand(C, H, W)
            else:
                a, b = rs.randn(C, H, W), rs.randn(C, H, W)
            a, b = a.astype(dt), b.astype(dt)
            v = ext.xcorrvol_cpu(t(a), t(b), D, bs).numpy()
            out["in0_%d" % k], out["in1_%d" % k], out["vol_%d" % k] = a, b, v
            out["argmax_%d" % k] = torch.argmax(t(v), 0).numpy()
            cases.append((C, H, W, D, bs))
            k += 1
    # tie case (SURVEY 8c fixture 6): constant columns in in1 force exact ties
    a = rs.rand(1, 14, 36).astype("float32")
    b = np.repeat(rs.rand(1, 14, 1), 36, axis=2).astype("float32")
    b[:, :, 20:] = rs.rand(1, 14, 16)
    v = ext.xcorrvol_cpu(t(a), t(b), 12, 9).numpy()
    out["in0_%d" % k], out["in1_%d" % k], out["vol_%d" % k] = a, b, v
    out["argmax_%d" % k] = torch.argmax(t(v), 0).numpy()
    cases.append((1, 14, 36, 12, 9))
    out["cases"] = np.array(cases, np.int64)
    save("xcorrvol_small", **out)


def kinect_crop(H=432, W=512):
    from PIL import Image
    im = np.array(Image.open("/root/reference/data/kinect_pattern.png").convert("L"))
    r0 = (im.shape[0] - H) // 2        # data/commons.py:22
    c0 = (im.shape[1] - W) // 2        # data/commons.py:23
    return np.ascontiguousarray(im[r0:r0 + H, c0:c0 + W])


def gen_cfg1(ext):
    """512x432x128 volumes of the reference on regenerable inputs:
    (a) uniform-random frames (seeds 1234 / 42), (b) the kinect pattern crop with a synthetic
    IR frame, both LCN'd by the repo's oracle LCN (so the inputs can be rebuilt anywhere)."""
    from oracle import oracle
    from tests import workloads
    H, W, D, bs = 432, 512, 128, 9
    sample = np.sort(np.random.RandomState(99).choice(D * H * W, 4096, replace=False))
    out = {"sample_idx": sample}
    a = workloads.uniform_frame(1234, H, W)
    b = workloads.uniform_frame(42, H, W)
    v = ext.xcorrvol_cpu(t(a), t(b), D, bs).numpy()
    out["uni_sha256"] = np.frombuffer(hashlib.sha256(v.tobytes()).digest(), np.uint8)
    out["uni_sample_val"] = v.reshape(-1)[sample]
    out["uni_argmax"] = torch.argmax(t(v), 0).numpy().astype(np.uint8)
    pat_u8 = kinect_crop(H, W)
    pat = pat_u8.astype(np.float32) / 255
    ir, disp = workloads.synth_ir(pat, np.random.RandomState(2024), D)
    ir_l, _ = oracle.lcn(ir[None, None], 5, 0.05)
    pat_l, _ = oracle.lcn(pat[None, None], 5, 0.05)
    v = ext.xcorrvol_cpu(t(ir_l[0]), t(pat_l[0]), D, bs).numpy()
    am = torch.argmax(t(v), 0).numpy()
    out["kin_pattern_u8"] = pat_u8
    out["kin_disp_gt"] = disp.astype(np.uint8)
    out["kin_inputs_sha256"] = np.frombuffer(hashlib.sha256(ir_l.tobytes() + pat_l.tobytes()).digest(), np.uint8)
    out["kin_sha256"] = np.frombuffer(hashlib.sha256(v.tobytes()).digest(), np.uint8)
    out["kin_sample_val"] = v.reshape(-1)[sample]
    out["kin_argmax"] = am.astype(np.uint8)
    crop = (slice(13, H - 13), slice(140, W - 13))          # exp_synph.py:46-50
    print("kinect cfg1: disparity MAE of reference argmax vs ground truth on eval crop: %.4f px"
          % np.abs(am[crop] - disp[crop]).mean())
    save("xcorrvol_cfg1", **out)


def gen_photometric(ext, te):
    out = {}
    rs = np.random.RandomState(11)
    cases = []
    k = 0
    for dt in ("float32", "float64"):
        for (B, C, H, W, bs) in [(2, 1, 24, 32, 9), (2, 3, 17, 19, 3), (1, 1, 10, 70, 9)]:
            es = rs.rand(B, C, H, W).astype(dt)
            ta = rs.rand(B, C, H, W).astype(dt)
            go = rs.randn(B, 1, H, W).astype(dt)
            out["es_%d" % k], out["ta_%d" % k], out["go_%d" % k] = es, ta, go
            for ty in range(4):
                for eps in (0.1, 0.5):
                    f = ext.photometric_loss_forward(t(es), t(ta), bs, ty, eps).numpy()
                    g = ext.photometric_loss_backward(t(es), t(ta), t(go), bs, ty, eps).numpy()
                    out["fwd_%d_%d_%g" % (k, ty, eps)] = f
                    out["bwd_%d_%d_%g" % (k, ty, eps)] = g
            cases.append((B, C, H, W, bs))
            k += 1
    out["cases"] = np.array(cases, np.int64)
    # the reference's own second opinion (functions.py:120-147) on case 0, f32
    es, ta = t(out["es_0"]), t(out["ta_0"])
    for name in ("mse", "sad", "census_mse", "census_sad"):
        out["pytorch_%s" % name] = te.photometric_loss_pytorch(es, ta, 9, name, 0.5).numpy()
    save("photometric", **out)


def gen_costvol(ext):
    """SAD / census volume by composition of reference ops."""
    rs = np.random.RandomState(5)
    H, W, D, bs = 20, 48, 16, 9
    im = rs.rand(H, W).astype(np.float32)
    pat = rs.rand(H, W).astype(np.float32)
    out = {"im": im, "pat": pat, "D": np.int64(D), "bs": np.int64(bs)}
    cols = np.arange(W)
    for ty in range(4):
        vol = np.empty((D, H, W), np.float32)
        for d in range(D):
            pd = np.ascontiguousarray(pat[:, np.clip(cols - d, 0, W - 1)])
            vol[d] = ext.photometric_loss_forward(t(pd)[None, None], t(im)[None, None], bs, ty, 0.5).numpy()[0, 0]
        out["vol_%d" % ty] = vol
        out["argmin_%d" % ty] = torch.argmin(t(vol), 0).numpy()
    save("costvol", **out)


def gen_lcn(nets):
    rs = np.random.RandomState(3)
    out = {}
    lcn = nets.LCN(5, 0.05)
    for k, (N, H, W, kind) in enumerate([(2, 24, 32, "u"), (1, 40, 53, "n"), (1, 64, 64, "flat")]):
        if kind == "u":
            x = rs.rand(N, 1, H, W)
        elif kind == "n":
            x = rs.randn(N, 1, H, W) * 3 + 10
        else:
            x = rs.rand(N, 1, H, W)
            x[:, :, 16:48, 16:48] = 1.0          # saturated flat block
        x = x.astype(np.float32)
        with torch.no_grad():
            y, s = lcn(t(x))
        out["x_%d" % k], out["y_%d" % k], out["std_%d" % k] = x, y.numpy(), s.numpy()
    lcn2 = nets.LCN(2, 0.1)
    x = rs.rand(1, 1, 13, 17).astype(np.float32)
    with torch.no_grad():
        y, s = lcn2(t(x))
    out["x_r2"], out["y_r2"], out["std_r2"] = x, y.numpy(), s.numpy()
    save("lcn_networks", **out)


def gen_lcn_cython():
    """data/lcn/lcn.pyx built under /tmp from where it lies (no copy into the repo)."""
    tmp = tempfile.mkdtemp(prefix="ctd_lcn_")
    setup = os.path.join(tmp, "setup.py")
    with open(setup, "w") as f:
        f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
                "setup(ext_modules=cythonize([Extension('lcn', ['/root/reference/data/lcn/lcn.pyx'])],"
                " build_dir=%r), include_dirs=[numpy.get_include()])\n" % tmp)
    try:
        subprocess.check_call([sys.executable, setup, "build_ext", "--build-lib", tmp, "--build-temp", tmp],
                              cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception as e:  # pragma: no cover
        print("cython LCN not built:", e)
        return
    sys.path.insert(0, tmp)
    import lcn as cy
    rs = np.random.RandomState(4)
    img = rs.rand(30, 41).astype(np.float32)
    y, s = cy.normalize(img, 5, 0.05)
    y2, s2 = cy.normalize(img, 2, 0.1)
    save("lcn_datagen", img=img, y_5=np.asarray(y), std_5=np.asarray(s), y_2=np.asarray(y2), std_2=np.asarray(s2))


def small_camera(H, W):
    K = np.array([[0.9 * W, 0, W / 2 - 0.3], [0, 0.92 * W, H / 2 + 0.2], [0, 0, 1]], np.float32)
    return K, np.linalg.inv(K).astype(np.float32)


def small_pose(rs, B):
    Rs, ts = [], []
    for _ in range(B):
        ax = rs.randn(3) * 0.02
        th = np.linalg.norm(ax)
        k = ax / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        Rs.append(R)
        ts.append(rs.randn(3) * 0.03)
    return np.stack(Rs).astype(np.float32), np.stack(ts).astype(np.float32)


def gen_losses(nets):
    rs = np.random.RandomState(21)
    B, H, W = 2, 24, 32
    out = {}
    # --- DispToDepth (networks.py:313-321) fwd + grad
    disp = (rs.rand(B, 1, H, W) * 40 - 4).astype(np.float32)     # some negatives -> relu
    bf = 0.075 * 567.6
    d2d = nets.DispToDepth(567.6, 0.075)
    dt = t(disp).requires_grad_(True)
    depth = d2d(dt)
    gd = rs.randn(*depth.shape).astype(np.float32)
    depth.backward(t(gd))
    out.update(d2d_disp=disp, d2d_depth=depth.detach().numpy(), d2d_go=gd, d2d_grad=dt.grad.numpy(),
               d2d_bf=np.float64(bf))
    # --- Sobel + DisparityLoss (networks.py:380-412, 537-565) fwd + grads, both branches
    disp = (np.cumsum(rs.rand(B, 1, H, W), axis=3) * 2).astype(np.float32)
    disp[:, :, 8:16, 10:20] += 7
    edge = rs.rand(B, 1, H, W).astype(np.float32)
    dl = nets.DisparityLoss()
    sob = nets.SobelFilter(norm=False)
    with torch.no_grad():
        out["sobel_grad"] = sob(t(disp)).numpy()
    dt, et = t(disp).requires_grad_(True), t(edge).requires_grad_(True)
    v = dl(dt, et)
    v.backward()
    out.update(dl_disp=disp, dl_edge=edge, dl_val=v.detach().numpy(), dl_gdisp=dt.grad.numpy(),
               dl_gedge=et.grad.numpy())
    dt = t(disp).requires_grad_(True)
    v = dl(dt)
    v.backward()
    out.update(dl_noedge_val=v.detach().numpy(), dl_noedge_gdisp=dt.grad.numpy())
    # --- geometric loss (networks.py:416-503) fwd + grads
    K, Ki = small_camera(H, W)
    R0, t0 = small_pose(rs, B)
    R1, t1 = small_pose(rs, B)
    depth0 = (1.5 + rs.rand(B, 1, H, W)).astype(np.float32)
    depth1 = (1.5 + rs.rand(B, 1, H, W)).astype(np.float32)
    for clamp in (0.1, -1.0):
        ge = nets.ProjectionDepthSimilarityLoss(t(K), t(Ki), H, W, clamp=clamp)
        a, b = t(depth0).requires_grad_(True), t(depth1).requires_grad_(True)
        v = ge(a, b, t(R0), t(t0), t(R1), t(t1))
        v.backward()
        tag = "c" if clamp > 0 else "nc"
        out.update({"ge_%s_val" % tag: v.detach().numpy(), "ge_%s_g0" % tag: a.grad.numpy(),
                    "ge_%s_g1" % tag: b.grad.numpy()})
        with torch.no_grad():
            out["ge_%s_fwd01" % tag] = ge.fwd(t(depth0), t(depth1), t(R0), t(t0), t(R1), t(t1)).numpy()
    base = nets.ProjectionBaseLoss(t(K), t(Ki), H, W)
    with torch.no_grad():
        uv, d = base(t(depth0), t(R0), t(t0), t(R1), t(t1))
    out.update(ge_K=K, ge_Ki=Ki, ge_R0=R0, ge_t0=t0, ge_R1=R1, ge_t1=t1, ge_depth0=depth0, ge_depth1=depth1,
               ge_uv=uv.numpy(), ge_d=d.numpy(), ge_ray=base.ray.numpy())
    save("losses", **out)


def gen_pattern_loss(nets):
    """RectifiedPatternSimilarityLoss (networks.py:340-378): loss, pattern_proj, dloss/ddisp."""
    rs = np.random.RandomState(31)
    B, H, W = 2, 24, 40
    pat = rs.randn(1, 3, H, W).astype(np.float32)
    disp = (rs.rand(B, 1, H, W) * 12).astype(np.float32)
    im = rs.randn(B, 1, H, W).astype(np.float32)
    std = (0.05 + rs.rand(B, 1, H, W)).astype(np.float32)
    out = dict(pattern=pat, disp=disp, im=im, std=std)
    for name in ("census_sad", "mse"):
        mod = nets.RectifiedPatternSimilarityLoss(H, W, t(pat), loss_type=name, loss_eps=0.5)
        for use_std in (True, False):
            dt = t(disp).requires_grad_(True)
            v, proj = mod(dt, t(im), t(std) if use_std else None)
            v.backward()
            tag = "%s_%d" % (name, use_std)
            out["val_" + tag], out["proj_" + tag], out["gdisp_" + tag] = v.detach().numpy(), proj.detach().numpy(), dt.grad.numpy()
    save("pattern_loss", **out)


def gen_nn_ops(ext):
    """nn_cpu / crosscheck_cpu / proj_nn_cpu of the compiled reference (ext_cpu.cpp:14-86)"""
    out = {}
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        a, b = workloads.nn_case(3, dt)
        i01 = ext.nn_cpu(t(a), t(b)).numpy()
        i10 = ext.nn_cpu(t(b), t(a)).numpy()
        out["nn01_" + tag], out["nn10_" + tag] = i01, i10
        out["cc_" + tag] = ext.crosscheck_cpu(t(i01), t(i10)).numpy()
        xyz0, xyz1, K = workloads.proj_case(4, dt)
        for ps in (1, 3, 4, 5):
            out["proj%d_%s" % (ps, tag)] = ext.proj_nn_cpu(t(xyz0), t(xyz1), t(K), ps).numpy()
    save("nn_ops", **out)


def gen_render():
    """RendererCpu<float>::render_mesh_proj of the reference (renderer/render/render_cpu.cpp:17-20) through
    oracle/ref_render_driver.cpp: depth / projected pattern / shaded ambient image of three small scenes"""
    from oracle import oracle as ora
    ref = build_ref.load_render().ctd_ref_render_mesh_proj
    out = {}
    for k, (seed, wall, shader) in enumerate(((1, True, (0.5, 1.5, 0.0, 10.0)), (2, True, (0.3, 1.0, 0.4, 8.0)),
                                              (3, False, (0.5, 1.5, 0.0, 10.0)))):
        sc = workloads.render_scene(seed, wall=wall)
        sc["shader"] = shader
        d, c, n = ora.render_mesh_proj(**sc, fn=ref)
        out["depth_%d" % k], out["color_%d" % k], out["normal_%d" % k] = d, c, n
        # RendererCpu<float>::render_mesh (render_cpu.cpp:12-15): the plain renderer on the same scenes
        d, c, n = ora.render_mesh(normals=workloads.render_normals(sc, seed), fn=build_ref.load_render().ctd_ref_render_mesh,
                                  **sc)
        out["mesh_depth_%d" % k], out["mesh_color_%d" % k], out["mesh_normal_%d" % k] = d, c, n
    save("render", **out)


def main():
    ext = build_ref.load()
    te, nets = ref_python.load()
    which = sys.argv[1:] or ["xcorr", "photo", "costvol", "lcn", "lcncy", "losses", "patloss", "nnops", "render", "cfg1"]
    if "xcorr" in which:
        gen_xcorrvol(ext)
    if "photo" in which:
        gen_photometric(ext, te)
    if "costvol" in which:
        gen_costvol(ext)
    if "lcn" in which:
        gen_lcn(nets)
    if "lcncy" in which:
        gen_lcn_cython()
    if "losses" in which:
        gen_losses(nets)
    if "patloss" in which:
        gen_pattern_loss(nets)
    if "nnops" in which:
        gen_nn_ops(ext)
    if "render" in which:
        gen_render()
    if "cfg1" in which:
        gen_cfg1(ext)


if __name__ == "__main__":
    main()
