"""Frame sharding across the GPUs of one node.

The hot path never mixes frames (every op is per frame, ext.h:220-222,235), so frames shard
with no data-path collective; the pattern and camera constants are replicated.  The only
exchange is the reduction of scalar losses: a ratio of sums such as the masked photometric
loss `(mask*diff).sum() / mask.sum()` (model/networks.py:377) must reduce numerator and
denominator separately -- the mean of per-rank ratios is not the batch value.

One process per GPU, `torch.distributed` ("nccl" is RCCL over xGMI on ROCm; "gloo" in CPU tests).
"""
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def check_gpu_count(n_ranks, backend, n_devices, what="--gpus"):
    """N ranks over RCCL need N devices: two ranks on one device fail late ("duplicate GPU") or hang.  The gloo
    rehearsal (CTD_DIST_BACKEND=gloo) may oversubscribe a device on purpose.  Returns an error text or None."""
    if backend == "nccl" and n_ranks > n_devices:
        return ("%s %d over RCCL needs %d visible GPUs, this box has %d (rehearse more ranks than GPUs with "
                "CTD_DIST_BACKEND=gloo)" % (what, n_ranks, n_ranks, n_devices))
    return None


def _parse_visible(value, total):
    """Number of devices a HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES style list leaves of `total` (the runtime stops
    at the first entry it cannot resolve; UUID entries are counted as given)."""
    n = 0
    for tok in value.split(","):
        tok = tok.strip()
        if not tok:
            break
        if tok.lstrip("-").isdigit():
            if not 0 <= int(tok) < total:
                break
        n += 1
    return n


def visible_gpu_count(kfd_nodes="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process tree would see, WITHOUT loading the HIP runtime: the launcher parent stays a process that never
    touched a GPU.  (`torch.cuda.device_count()` falls back to `hipGetDeviceCount` whenever amdsmi cannot be imported,
    which initialises the runtime in the caller.)  KFD topology nodes with a non-zero `simd_count` are the GPUs (CPU
    nodes report 0); ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES, narrow the list the way the
    runtime applies them.  Returns -1 when the topology cannot be read (no driver: the caller decides)."""
    try:
        nodes = sorted(os.listdir(kfd_nodes), key=lambda s: int(s) if s.isdigit() else 1 << 30)
    except OSError:
        return -1
    total = 0
    for node in nodes:
        try:
            with open(os.path.join(kfd_nodes, node, "properties")) as fh:
                props = dict(line.split(None, 1) for line in fh if len(line.split(None, 1)) == 2)
        except OSError:
            continue                                    # a node of another container's cgroup: not ours to use
        if int(props.get("simd_count", "0").strip() or 0) > 0:
            total += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in os.environ:
            total = _parse_visible(os.environ[var], total)
            if var != "ROCR_VISIBLE_DEVICES":
                break                                   # HIP_ and CUDA_VISIBLE_DEVICES are one setting (HIP's wins)
    return total


def _device_count_in_child(timeout_s=120):
    """torch.cuda.device_count() of a fresh interpreter (-1 when it cannot be had)."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                             timeout=timeout_s)
        return int(out.stdout.decode().strip().splitlines()[-1])
    except Exception:                                   # noqa: BLE001
        return -1


def launch_ranks(script, argv, n_ranks, deadline_s=None, capture_rank0=True):
    """Start `n_ranks` fresh interpreters of `script` (one process per GPU; this parent does not load the HIP runtime
    -- it counts GPUs from the KFD topology -- and never re-execs), rendezvous on 127.0.0.1, relay rank 0's stdout while
    the ranks run, return an exit code.  A rank that dies takes the others down with it, and the whole launch has a
    deadline (CTD_BENCH_DEADLINE_S, default 900 s): no wedged rank keeps the parent waiting."""
    import socket
    import threading
    backend = os.environ.get("CTD_DIST_BACKEND", "nccl")
    n_dev = visible_gpu_count()
    if backend == "nccl" and n_ranks > 1:
        # the topology lists every GPU of the host even where a device cgroup lets this container open fewer, and it may
        # be unreadable altogether: the count that decides is the runtime's -- asked in a short-lived CHILD, so that this
        # parent still never loads HIP.  (One rank needs no second opinion.)
        n_child = _device_count_in_child()
        if n_child >= 0:
            n_dev = n_child if n_dev < n_ranks else min(n_dev, n_child)
    err = check_gpu_count(n_ranks, backend, n_dev)
    if err:
        sys.stderr.write("%s: %s\n" % (os.path.basename(script), err))
        return 2
    if deadline_s is None:
        deadline_s = float(os.environ.get("CTD_BENCH_DEADLINE_S", "900"))
    # the port is bound here and released right before the ranks start; a rendezvous that loses the race fails fast
    # (rank 0 cannot bind) and the loop below then ends the other ranks
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=(subprocess.PIPE if capture_rank0 else None) if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained while the ranks run: a rank that prints more than a pipe buffer (verbose RCCL logs)
    # would otherwise block in write() until the deadline
    chunks = []

    def _drain(fh):
        for block in iter(lambda: fh.read(65536), b""):
            chunks.append(block)

    reader = None
    if procs[0].stdout is not None:
        reader = threading.Thread(target=_drain, args=(procs[0].stdout,), daemon=True)
        reader.start()
    t_end = time.monotonic() + deadline_s
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or time.monotonic() > t_end:
            failed = ("rank %d exited with code %d" % (bad[0], rcs[bad[0]])) if bad else "deadline of %.0f s passed" % deadline_s
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    if reader is not None:
        reader.join(timeout=10)
    out = b"".join(chunks)
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if failed:
        sys.stderr.write("%s: %s; the remaining ranks were stopped\n" % (os.path.basename(script), failed))
        return 1
    return max(abs(p.returncode) for p in procs)


def frame_shard(n_frames, rank=None, world_size=None):
    """Contiguous, balanced [begin, end) slice of `n_frames` for this rank (first ranks take the remainder).
    For frame pairs of a track (geometric loss) shard on the batch axis so both frames of a pair stay together."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, rem = divmod(int(n_frames), world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_ratio(numerator, denominator, group=None):
    """Global `sum(numerator) / sum(denominator)` over all ranks: one all-reduce of two scalars
    (8 bytes over xGMI, latency-bound).  Differentiable w.r.t. the local numerator / denominator."""
    local = torch.stack([numerator.reshape(()), denominator.reshape(())])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        total = local.detach().clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        # straight-through: value of the global sums, gradient of the local contribution
        total = local + (total - local.detach())
    else:
        total = local
    return total[0] / total[1]


def reduce_ratio_ddp(numerator, denominator, group=None):
    """The same global ratio for a loss term of a DistributedDataParallel step.  DDP AVERAGES the ranks' gradients, so
    a term whose batch value is `sum_r(num_r) / sum_r(den_r)` (model/networks.py:377: the mask-weighted photometric
    mean; also any mean over a per-rank varying number of samples) must hand DDP the local gradient
    `world * d num_r / DEN`: averaged over the ranks that is `sum_r(d num_r) / DEN`, the gradient of the reference's
    single-process batch.  Value: the global ratio on every rank (what the reference logs); the denominator carries
    no gradient (in the reference it is the LCN std of the input, or a sample count)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        # one process: the reference's plain division (networks.py:377), bit for bit where the denominator is positive --
        # a multiply by 1 / d rounds differently -- and a zero term (value and gradient) where it is not
        d1 = denominator.reshape(()).detach()
        ok = d1 > 0
        return numerator.reshape(()) / torch.where(ok, d1, torch.ones_like(d1)) * ok.to(d1.dtype)
    world = dist.get_world_size(group)
    local = torch.stack([numerator.reshape(()), denominator.reshape(())]).detach()
    total = local.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    total = total.to(numerator.device)
    # A global denominator of 0 (no supervised sample on any rank: the reference supervises only the last 256 ids,
    # exp_synph.py:39) is a term of value 0 and gradient 0 -- never `world / tiny`, which is inf in f32 from world = 4
    # and turns the zero numerator into NaN.
    ok = total[1] > 0
    den = torch.where(ok, total[1], torch.ones_like(total[1]))
    scale = torch.where(ok, float(world) / den, torch.zeros_like(den))
    grad_path = numerator.reshape(()) * scale
    value = torch.where(ok, total[0] / den, torch.zeros_like(den))
    return grad_path + (value - grad_path.detach())


def reduce_mean(value, count, group=None):
    """Global mean of per-rank means over unequal shard sizes: sum(value*count) / sum(count)."""
    c = torch.as_tensor(float(count), dtype=value.dtype, device=value.device)
    return reduce_ratio(value * c, c, group)


def gather_scalars(value, group=None):
    """All-gather of one scalar per rank (the north_star's loss all-gather); returns a [world] tensor."""
    v = value.detach().reshape(1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        out = [torch.empty_like(v) for _ in range(dist.get_world_size(group))]
        dist.all_gather(out, v, group=group)
        return torch.cat(out)
    return v


def gather_scalars_async(value, out=None, group=None, force=False):
    """The same exchange without stalling the compute stream: the all-gather is queued on the collective's own stream
    behind `value` and the caller's stream does not wait for it.  Returns (out [world], work); `work` is None when
    there is nothing to exchange.  Keep `out` alive and call `work.wait()` (a stream-level wait on RCCL, no host block)
    before reading or reusing it -- in a steady loop that is one or more steps later, when the exchange has long
    finished, so the 4 bytes per rank and the skew between ranks never sit on the step's critical path.
    `force`: go through the collective even in a one-rank group (the RCCL rehearsal on a one-GPU box)."""
    v = value.detach().reshape(1).contiguous()
    if not (dist.is_initialized() and (force or dist.get_world_size(group) > 1)):
        return v, None
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(world, dtype=v.dtype, device=v.device)
    work = dist.all_gather_into_tensor(out, v, group=group, async_op=True)
    return out, work
