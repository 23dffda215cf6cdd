#!/usr/bin/env python3
"""Synthetic IR frames on the GPU box, the way data/create_syn_data.py:152-188 makes them: render the scene with
the dot pattern projected from 7.5 cm to the side, blend the reflected pattern with the shaded ambient image,
derive the ground-truth disparity from the depth buffer.

    python examples/render_synthetic.py [--height 480 --width 640 --boxes 200]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--boxes", type=int, default=200)
    ap.add_argument("--frames", type=int, default=4)
    args = ap.parse_args()
    from connecting_the_dots_amd import renderer
    from tests import workloads
    H, W = args.height, args.width
    baseline = 0.075
    pat01 = workloads.syn_dot_pattern(H, W, seed=42)
    pattern = np.repeat(pat01[:, :, None], 3, axis=2).astype(np.float32)
    for i in range(args.frames):
        sc = workloads.render_scene(100 + i, H=H, W=W, n_boxes=args.boxes)
        K, R, t, _, _ = sc["cam"]
        cam = renderer.PyCamera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], R, t, W, H)
        proj = renderer.PyCamera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], R, np.array([baseline, 0, 0], np.float32), W, H)
        data = renderer.PyRenderInput(verts=sc["verts"], colors=sc["colors"], faces=sc["faces"])
        r = renderer.PyRenderer(cam, renderer.PyShader(0.5, 1.5, 0.0, 10), engine='gpu')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.mesh_proj(data, proj, pattern, d_alpha=0, d_beta=0.35)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        im, depth, ambient = r.color().mean(2), r.depth(), r.normal().mean(2)
        disp = baseline * K[0, 0] / depth                                    # create_syn_data.py:163
        blend = 0.6
        ir = blend * im + (1 - blend) * ambient                              # create_syn_data.py:171
        print("frame %d: %d faces, %.1f ms, %.1f Mray*tri/s, lit %.1f %%, disp %.1f..%.1f px, IR mean %.3f" % (
            i, len(sc["faces"]), dt * 1e3, 2 * H * W * len(sc["faces"]) / dt / 1e6, 100 * (im > 0).mean(),
            disp[depth > 0].min(), disp[depth > 0].max(), ir.mean()), flush=True)


if __name__ == "__main__":
    main()
