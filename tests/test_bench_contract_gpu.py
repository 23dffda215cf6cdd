"""bench.py's one-line JSON contract (metric / value / roofline / cpu_baseline ...) on a short run, N = 1 and the
self-launched N = 2 rehearsal over gloo (two ranks sharing the box's one GPU; RCCL itself needs one GPU per rank).
Checks the shape of the line and the arithmetic between its fields, not the speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=e,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    j = run_bench("--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["unit"] == "Mpix*disp/s" and j["dtype"] == "f32" and j["data"] == "synthetic" and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["disparity_mae_vs_ref"] == 0.0
    cfg = j["config"]
    assert "config 2" in cfg["workload"] and cfg["frames_per_gpu"] == 16 and (cfg["H"], cfg["W"], cfg["D"]) == (432, 512, 128)
    # value = units of all ranks / wall time of the timed region
    units = cfg["frames_per_gpu"] * cfg["H"] * cfg["W"] * cfg["D"]
    assert abs(j["value"] - units / (j["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["launches"] == 4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.2 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    assert len(j["repeat_ms_per_step"]) == 2 and j["settle_steps"] >= 20
    assert {"volume_kernel_alone", "fused_volume_free"} <= set(j["also_measured"])
    # the second region (config 3's step) at the same N, like for like with the N > 1 lines
    c3 = j["config3"]
    assert "config 3" in c3["workload"] and c3["steps"] == 4 and c3["ms_per_step"] > j["ms_per_step"] * 0.9
    assert abs(c3["value"] - units / (c3["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * c3["value"]
    assert len(c3["loss_allgather"]) == 1 and c3["loss_allgather"][0] == c3["loss_allgather"][0]
    # end-to-end index parity of the timed step from the raw frame through the exact LCN: reported, near-ties only
    e2e = j["disparity_mae_detail"]["timed_workload_frame0_vs_exact_lcn_pipeline"]
    assert e2e["pixels"] == 432 * 512 and e2e["pixels_differing"] <= 8 and "scaling_note" not in cfg


def test_two_ranks_self_launched_over_gloo():
    j = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", env={"CTD_DIST_BACKEND": "gloo"})
    # the headline is the SAME config-2 step as at N = 1 (no collective in it); config 3 is the second block
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and "config 2" in j["config"]["workload"]
    units = 2 * 16 * 432 * 512 * 128
    assert abs(j["value"] - units / (j["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * j["value"]
    c3 = j["config3"]
    assert "config 3" in c3["workload"] and "gloo" in c3["parallelism"]
    assert len(c3["loss_allgather"]) == 2 and all(v == v for v in c3["loss_allgather"])
    assert abs(c3["value"] - units / (c3["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * c3["value"]
