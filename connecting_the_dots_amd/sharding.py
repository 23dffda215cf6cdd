"""Frame sharding across the GPUs of one node.

The hot path never mixes frames (every op is per frame, ext.h:220-222,235), so frames shard
with no data-path collective; the pattern and camera constants are replicated.  The only
exchange is the reduction of scalar losses: a ratio of sums such as the masked photometric
loss `(mask*diff).sum() / mask.sum()` (model/networks.py:377) must reduce numerator and
denominator separately -- the mean of per-rank ratios is not the batch value.

One process per GPU, `torch.distributed` ("nccl" is RCCL over xGMI on ROCm; "gloo" in CPU tests).
"""
import torch
import torch.distributed as dist


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


def gather_scalars_async(value, out=None, group=None):
    """The same exchange without stalling the compute stream: the all-gather is queued on the collective's own stream
    behind `value` and the caller's stream does not wait for it.  Returns (out [world], work); `work` is None when
    there is nothing to exchange.  Keep `out` alive and call `work.wait()` (a stream-level wait on RCCL, no host block)
    before reading or reusing it -- in a steady loop that is one or more steps later, when the exchange has long
    finished, so the 4 bytes per rank and the skew between ranks never sit on the step's critical path."""
    v = value.detach().reshape(1).contiguous()
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return v, None
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(world, dtype=v.dtype, device=v.device)
    work = dist.all_gather_into_tensor(out, v, group=group, async_op=True)
    return out, work
