#!/usr/bin/env python3
"""Config 5 stand-in (no ShapeNet / renderer offline): train the small disparity network on synthetic IR frames
(dot pattern shifted by a seeded piecewise-constant disparity map, tests/workloads.synth_ir) with the HIP losses.

    python examples/train_synthetic.py [--iters 50] [--batch 8] [--algo fast|exact]

Prints it/s and the per-bucket timings of the reference's StopWatch names.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_batch(pattern01, rs, B, D):
    from tests import workloads
    frames = [workloads.synth_ir(pattern01, rs, D)[0] for _ in range(B)]
    return torch.from_numpy(np.stack(frames)[:, None].astype(np.float32))


def make_rendered_batch(pattern01, seed, B, n_boxes=60):
    """IR frames from the HIP ray caster the way data/create_syn_data.py:152-171 composes them: reflected dot
    pattern (projector 7.5 cm to the side, decay 0.35) blended 0.6 / 0.4 with the Phong-shaded ambient image"""
    from connecting_the_dots_amd import renderer
    from tests import workloads
    H, W = pattern01.shape
    pattern = np.repeat(pattern01[:, :, None], 3, axis=2).astype(np.float32)
    frames = []
    for b in range(B):
        sc = workloads.render_scene(seed * 1000 + b, H=H, W=W, n_boxes=n_boxes)
        K, R, t, _, _ = sc["cam"]
        cam = renderer.PyCamera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], R, t, W, H)
        proj = renderer.PyCamera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], R, np.array([0.075, 0, 0], np.float32), W, H)
        r = renderer.PyRenderer(cam, renderer.PyShader(0.5, 1.5, 0.0, 10), engine='gpu')
        r.mesh_proj(renderer.PyRenderInput(verts=sc["verts"], colors=sc["colors"], faces=sc["faces"]), proj, pattern,
                    d_alpha=0, d_beta=0.35)
        frames.append(0.6 * r.color().mean(2) + 0.4 * r.normal().mean(2))
    return torch.from_numpy(np.stack(frames)[:, None].astype(np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=5, help="iterations excluded from the timings (MIOpen kernel search)")
    ap.add_argument("--algo", default="fast", choices=["fast", "exact"])
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--rendered", action="store_true", help="frames from the HIP renderer instead of shifted patterns")
    args = ap.parse_args()
    from connecting_the_dots_amd import torchext as te
    from connecting_the_dots_amd.train import DisparityTrainer, SmallDispEdgeNet
    from tests import workloads
    dev = torch.device("cuda", 0)
    H, W, D = args.height, args.width, 64
    torch.manual_seed(0)
    rs = np.random.RandomState(0)
    pattern01 = workloads.syn_dot_pattern(H, W, seed=42)
    pat_lcn, _ = te.lcn(torch.from_numpy(pattern01[None, None]).to(dev), 5, 0.05)
    trainer = DisparityTrainer(SmallDispEdgeNet(max_disp=D), pat_lcn, H, W, algo=args.algo)
    if args.rendered:
        data = [make_rendered_batch(pattern01, k, args.batch) for k in range(4)]
    else:
        data = [make_batch(pattern01, rs, args.batch, D) for _ in range(4)]
    from connecting_the_dots_amd.train import StopWatch
    for it in range(args.iters):
        if it == args.warmup:
            trainer.watch = StopWatch(dev)
        ir = data[it % len(data)].to(dev, non_blocking=True)           # copy_data
        vals = trainer.train_step(ir)
        if it % 10 == 0 or it == args.iters - 1:
            print("iter %3d  loss %s" % (it, " ".join("%.5f" % v for v in vals)), flush=True)
    ms = trainer.watch.mean_ms()
    print("it/s %.1f | ms per step: %s" % (1e3 / ms["total"], ", ".join("%s %.2f" % kv for kv in ms.items())))


if __name__ == "__main__":
    main()
