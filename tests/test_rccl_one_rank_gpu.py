"""RCCL on the box we have: ONE rank, backend "nccl" (= RCCL on ROCm).  Every collective of the repo had only ever run
over gloo; the first RCCL call must not be the driver's 8-GPU run.  (i) bench.py's N > 1 code path (config 3:
init_process_group("nccl", device_id=...), the non-blocking scalar all-gather on a device tensor, the f64 all-reduce
(MAX) of the elapsed time, barrier, destroy_process_group) with `--force-dist`; (ii) two TrackTrainer steps under
DistributedDataParallel with the process group on nccl, against the same two steps without a process group.
Each runs in a child process (a process group per process; the parent pytest process stays without one).  Logs go to
gpurun_out/rccl_one_rank_*.log when that directory exists (copied to profiles/ by hand)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _save_log(name, text):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, name), "w") as fh:
            fh.write(text)


def _env():
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
             NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "VERSION"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "CTD_DIST_BACKEND"):
        e.pop(k, None)
    return e


def test_bench_n_gt_1_path_on_one_rccl_rank():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist",
                        "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=_env(),
                       timeout=600, cwd=ROOT)
    _save_log("rccl_one_rank_bench.log", p.stdout + "\n--- stderr ---\n" + p.stderr[-6000:])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(p.stdout.strip().splitlines()) == 1, "stdout carries more than the JSON line (RCCL's banner?): %r" % p.stdout[:300]
    j = json.loads(lines[0])
    cfg, c3 = j["config"], j["config3"]
    # both blocks: the config-2 headline (no collective) and config 3 with its all-gather over RCCL
    assert cfg["ranks_seen"] == 1 and "config 2" in cfg["workload"]
    assert "RCCL" in c3["parallelism"] and "config 3" in c3["workload"]
    assert len(c3["loss_allgather"]) == 1 and c3["loss_allgather"][0] == c3["loss_allgather"][0]      # gathered, finite
    assert j["n_gpus"] == 1 and j["disparity_mae_vs_ref"] == 0.0
    units = 16 * 432 * 512 * 128
    assert abs(j["value"] - units / (j["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * j["value"]
    assert abs(c3["value"] - units / (c3["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * c3["value"]


def test_track_trainer_ddp_on_one_rccl_rank(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               CTD_DDP_BACKEND="nccl")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    _save_log("rccl_one_rank_ddp.log", p.stdout + "\n--- stderr ---\n" + p.stderr[-6000:])
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    got = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    # the same two steps in this process without a process group: one rank over RCCL must change nothing but rounding
    from tests.test_config5_ddp_gpu import make_setup
    from connecting_the_dots_amd.train import TrackTrainer
    net, pats, K, batch = make_setup()
    tr = TrackTrainer(net, pats, K, 0.075, [567.6 / 4 / 2 ** s for s in range(4)], train_edge=0)
    vals = tr.train_step(batch)
    grads = [q.grad.detach().clone().cpu() for q in tr.net.parameters()]
    tr.train_step(batch)
    params = [q.detach().clone().cpu() for q in tr.net.parameters()]
    assert len(got["vals"]) == len(vals)
    for i, (a, b) in enumerate(zip(got["vals"], vals)):
        assert abs(a - b) <= 2e-5 * abs(b) + 1e-7, (i, a, b)
    gscale = max(float(g.abs().max()) for g in grads)
    for a, b in zip(got["grads"], grads):
        assert float((a - b).abs().max()) <= 2e-3 * gscale
    for a, b in zip(got["params"], params):
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 1e-3
