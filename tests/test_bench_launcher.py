"""bench.py's own launcher (`--gpus N` without torch.distributed.run), on the CPU: the GPU-count check fails at once
with a clear message, and a rank that dies takes the launch down instead of leaving the parent waiting."""
import importlib.util
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpu_count_check():
    b = _bench()
    assert b.check_gpu_count(1, "nccl", 1) is None
    assert b.check_gpu_count(8, "nccl", 8) is None
    msg = b.check_gpu_count(2, "nccl", 1)
    assert msg and "--gpus 2" in msg and "1" in msg
    assert b.check_gpu_count(2, "gloo", 1) is None          # the rehearsal may share a device


def test_self_launch_refuses_more_ranks_than_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        return                                               # a multi-GPU box would really start the ranks
    env = dict(os.environ, CTD_DIST_BACKEND="nccl")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, timeout=120)
    assert p.returncode == 2, (p.returncode, p.stderr.decode()[-400:])
    assert b"needs 2 visible GPUs" in p.stderr
    assert time.time() - t0 < 60


def test_self_launch_stops_the_other_ranks_when_one_dies():
    import torch
    if torch.cuda.is_available():
        return                                               # on a GPU box the ranks would run the bench
    # gloo rehearsal on a box without a GPU: every rank exits at once ("bench.py needs a GPU"); the parent must come
    # back non-zero well inside its deadline, not wait on anybody
    env = dict(os.environ, CTD_DIST_BACKEND="gloo", CTD_BENCH_DEADLINE_S="90")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, timeout=150)
    assert p.returncode != 0
    assert b"remaining ranks were stopped" in p.stderr or b"needs a GPU" in p.stderr
    assert time.time() - t0 < 80
