#!/usr/bin/env python3
"""BASELINE config 5: the multi-scale photometric + geometric training step at its real shape (480x640, batch 8 per
GPU, track length 2, four scales, the full-size disparity / edge network) on synthetic tracks, data parallel.

    python examples/train_config5.py [--gpus N] [--iters 30] [--batch 8]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_config5.py --gpus N

`--gpus N` without a launcher starts the N ranks itself (one process per GPU, DistributedDataParallel over RCCL; set
CTD_DIST_BACKEND=gloo to rehearse N ranks on fewer GPUs).  Rank 0 prints it/s of the whole job and the per-bucket
timings under the reference's StopWatch names (torchext/worker.py:362-443).  No dataset ships offline: frames are the
seeded dot pattern under piecewise-constant disparities (tests/workloads.track_batch), weights random-initialised.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5, help="iterations excluded from the timings (MIOpen kernel search)")
    ap.add_argument("--batch", type=int, default=8, help="tracks per GPU (reference: train_batch_size 8)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    # the measured setting of profiles/round3_config5.txt (18.2 against 15.75 it/s) is the default since round 4
    ap.add_argument("--miopen-find", action=argparse.BooleanOptionalAction, default=True,
                    help="torch.backends.cudnn.benchmark: MIOpen searches its kernels per shape (--no-miopen-find: heuristics)")
    ap.add_argument("--channels-last", action=argparse.BooleanOptionalAction, default=True,
                    help="network weights and activations in NHWC (--no-channels-last: NCHW)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # self-launch: this parent never touches a GPU
        from connecting_the_dots_amd import sharding
        sys.exit(sharding.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, capture_rank0=False))

    import numpy as np
    import torch
    import torch.nn.functional as F
    from connecting_the_dots_amd import torchext as te
    from connecting_the_dots_amd.nets import DispEdgeNet
    from connecting_the_dots_amd.train import StopWatch, TrackTrainer
    from tests import workloads
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    from connecting_the_dots_amd import sharding
    err = sharding.check_gpu_count(world, os.environ.get("CTD_DIST_BACKEND", "nccl"), torch.cuda.device_count())
    if err:
        raise SystemExit("train_config5.py: " + err)
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()     # (modulo: gloo rehearsal only)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("CTD_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        pg = dist.group.WORLD
    H, W, D, TL = args.height, args.width, 128, 2
    focal, baseline = 567.6, 0.075
    K = torch.tensor([[focal, 0, 324.7], [0, 570.2, 250.1], [0, 0, 1]], device=dev)
    batches = []
    for k in range(2):
        nb = workloads.track_batch(100 * rank + k, TL, args.batch, H, W, D)
        pat01 = nb.pop("pattern")
        batches.append({key: torch.from_numpy(v).to(dev) for key, v in nb.items()})
    pats, p = [], torch.from_numpy(pat01[None, None]).to(dev)
    for s in range(4):
        pats.append(te.lcn(p.contiguous(), 5, 0.05)[0])
        p = F.avg_pool2d(p, 2)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    net = DispEdgeNet(2, D)
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
    tr = TrackTrainer(net, pats, K, baseline, [focal / 2 ** s for s in range(4)], process_group=pg,
                      device_ids=[dev_index] if pg is not None else None)
    for it in range(args.iters):
        if it == args.warmup:
            tr.watch = StopWatch(dev)
        vals = tr.train_step(batches[it % len(batches)])
        if rank == 0 and (it % 10 == 0 or it == args.iters - 1):
            print("iter %3d  loss %.5f  (photo %s | disp %.4f | edge %s | geo %s)" % (
                it, sum(vals), " ".join("%.4f" % v for v in vals[:4]), vals[4], " ".join("%.4f" % v for v in vals[5:8]),
                " ".join("%.5f" % v for v in vals[8:])), flush=True)
    ms = tr.watch.mean_ms()
    if rank == 0:
        print("config 5: %d GPU(s) x batch %d x track %d at %dx%d: %.2f it/s, %.1f frames/s | ms per step: %s" % (
            world, args.batch, TL, H, W, 1e3 / ms["total"], world * args.batch * TL * 1e3 / ms["total"],
            ", ".join("%s %.2f" % kv for kv in ms.items())), flush=True)
    # the reference's test pass (exp_synph.py:201-224): disparity error of scale 0 on the evaluation crop.  Under DDP the
    # loss terms reduce across the ranks, so EVERY rank runs it (on its own shard); rank 0 prints its shard's metric.
    ev_vals, metric = tr.evaluate(batches[0])
    if rank == 0:
        print("evaluate: loss %.5f | %s" % (sum(ev_vals), ", ".join("%s=%.4f" % kv for kv in metric.items())), flush=True)
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
