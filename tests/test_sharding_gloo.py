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


def _worker_zero_den(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from connecting_the_dots_amd import sharding
    # the edge term of a config-5 step without a supervised sample on ANY rank (exp_synph.py:39 supervises the last 256
    # ids only): numerator 0 * something, count 0 -- value and gradient must be 0, not NaN (f32: world / tiny = inf
    # from world = 4)
    w = torch.ones(3, dtype=torch.float32, requires_grad=True)
    num = (w * 0.0).sum()
    cnt = torch.zeros((), dtype=torch.float32)
    val = sharding.reduce_ratio_ddp(num, cnt)
    val.backward()
    # and an ordinary step right after: one rank has samples, the others none
    w2 = torch.ones(3, dtype=torch.float32, requires_grad=True)
    num2 = (w2 * float(rank == 1)).sum()
    cnt2 = torch.tensor(3.0 if rank == 1 else 0.0)
    val2 = sharding.reduce_ratio_ddp(num2, cnt2)
    val2.backward()
    q.put((rank, float(val), w.grad.tolist(), float(val2), w2.grad.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_ratio_ddp_zero_global_denominator_world4():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 4
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_zero_den, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, v, g, v2, g2 in res:
        assert v == 0.0 and g == [0.0, 0.0, 0.0]                          # finite zeros, no NaN
        assert abs(v2 - 1.0) < 1e-6                                       # 3 / 3 on every rank
        # DDP averages: world * d num_r / DEN -> 4/3 per element on the rank that has the samples, 0 elsewhere
        assert all(abs(x - (4.0 / 3.0 if rank == 1 else 0.0)) < 1e-6 for x in g2)


def test_ratio_ddp_single_process_zero_denominator():
    from connecting_the_dots_amd import sharding
    w = torch.ones(2, requires_grad=True)
    v = sharding.reduce_ratio_ddp((w * 0.0).sum(), torch.zeros(()))
    v.backward()
    assert float(v.detach()) == 0.0 and w.grad.tolist() == [0.0, 0.0]


def test_visible_gpu_count_reads_kfd_without_the_runtime(tmp_path, monkeypatch):
    from connecting_the_dots_amd import sharding
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):                   # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\n" % (64 if simd == 0 else 0, simd))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert sharding.visible_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert sharding.visible_gpu_count(str(tmp_path)) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,7,0")                    # the runtime stops at the first bad entry
    assert sharding.visible_gpu_count(str(tmp_path)) == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert sharding.visible_gpu_count(str(tmp_path)) == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    assert sharding.visible_gpu_count(str(tmp_path)) == 1
    assert sharding.visible_gpu_count(str(tmp_path / "missing")) == -1


def test_launch_ranks_drains_a_chatty_rank0(tmp_path):
    """rank 0 printing far more than a pipe buffer must not block until the deadline."""
    from connecting_the_dots_amd import sharding
    script = tmp_path / "chatty.py"
    script.write_text("import os, sys\n"
                      "if os.environ['RANK'] == '0':\n"
                      "    sys.stdout.write('x' * (1 << 20)); sys.stdout.write('\\nDONE\\n')\n")
    import io
    import contextlib
    import time
    os.environ["CTD_DIST_BACKEND"] = "gloo"
    try:
        buf = io.StringIO()
        t0 = time.time()
        with contextlib.redirect_stdout(buf):
            rc = sharding.launch_ranks(str(script), [], 2, deadline_s=60)
        assert rc == 0 and time.time() - t0 < 30
        assert buf.getvalue().endswith("DONE\n") and len(buf.getvalue()) > (1 << 20)
    finally:
        os.environ.pop("CTD_DIST_BACKEND", None)
