"""One rank of tests/test_config5_ddp_gpu.py: TrackTrainer under DistributedDataParallel (gloo rendezvous, the HIP
losses on the one GPU of the box), two steps; saves the loss values of step 1, the gradients after step 1's backward
and the parameters after step 2.   python tests/ddp_worker.py OUT_DIR   (RANK / WORLD_SIZE / MASTER_* in the env)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.test_config5_ddp_gpu import make_setup, shard_batch          # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if os.environ.get("CTD_DDP_BACKEND", "gloo") == "nccl":             # RCCL (tests/test_rccl_one_rank_gpu.py: one rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from connecting_the_dots_amd.train import TrackTrainer
    net, pats, K, batch = make_setup()
    tr = TrackTrainer(net, pats, K, 0.075, [567.6 / 4 / 2 ** s for s in range(4)], train_edge=0,
                      process_group=dist.group.WORLD, device_ids=[0])
    mine = shard_batch(batch, rank, world)
    vals1 = tr.train_step(mine)
    grads = [p.grad.detach().clone().cpu() for p in tr.net.parameters()]
    tr.train_step(mine)
    params = [p.detach().clone().cpu() for p in tr.net.parameters()]
    torch.save({"vals": vals1, "grads": grads, "params": params}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
