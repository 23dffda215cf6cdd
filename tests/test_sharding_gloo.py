"""world_size-2 gloo test (CPU) of the frame sharding and the scalar loss reductions of SURVEY 8e."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from connecting_the_dots_amd import sharding
    torch.manual_seed(0)
    n = 7                                               # uneven on purpose
    diff = torch.rand(n, 4, 5, dtype=torch.float64)     # "photometric diff" per frame (same on all ranks)
    mask = torch.rand(n, 4, 5, dtype=torch.float64)
    b, e = sharding.frame_shard(n)
    d_local = diff[b:e].clone().requires_grad_(True)
    num = (mask[b:e] * d_local).sum()
    den = mask[b:e].sum()
    val = sharding.reduce_ratio(num, den)               # networks.py:377 across ranks
    val.backward()
    full = (mask * diff).sum() / mask.sum()
    grad_full = mask[b:e] / mask.sum()
    per_rank_mean = diff[b:e].mean()
    gm = sharding.reduce_mean(per_rank_mean, (e - b) * 20)
    gathered = sharding.gather_scalars(per_rank_mean)
    # the non-blocking form (bench.py's config-3 step): same values once waited for, into a caller-owned buffer
    buf = torch.full((world,), -1.0, dtype=per_rank_mean.dtype)
    out_async, work = sharding.gather_scalars_async(per_rank_mean, out=buf)
    work.wait()
    assert out_async is buf and torch.equal(buf, gathered)
    q.put((rank, (b, e), float(val), float(full), float((d_local.grad - grad_full).abs().max()), float(gm),
           float(diff.mean()), gathered.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_reduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, v0, f0, g0, m0, fm0, ga0), (r1, s1, v1, f1, g1, m1, fm1, ga1) = res
    assert s0 == (0, 4) and s1 == (4, 7)                                 # contiguous, balanced, complete
    assert abs(v0 - f0) < 1e-12 and abs(v1 - f1) < 1e-12 and v0 == v1    # ratio of global sums, same everywhere
    assert g0 < 1e-12 and g1 < 1e-12                                     # local gradient of the global ratio
    assert abs(m0 - fm0) < 1e-12 and abs(m1 - fm1) < 1e-12               # size-weighted mean
    assert ga0 == ga1 and len(ga0) == 2


def test_frame_shard_covers_everything():
    from connecting_the_dots_amd import sharding
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [sharding.frame_shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
