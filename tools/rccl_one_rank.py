import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
from connecting_the_dots_amd import sharding
v=torch.tensor(3.5, device=dev)
out, work = sharding.gather_scalars_async(v)
if work is not None: work.wait()
print("gather", out, work)
t=torch.tensor([1.0], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("allreduce", t)
print("ratio", sharding.reduce_ratio_ddp(torch.tensor(2.0, device=dev, requires_grad=True), torch.tensor(4.0, device=dev)))
dist.barrier(); dist.destroy_process_group(); print("ok")
