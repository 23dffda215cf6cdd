"""Minimal training step of the disparity network on the HIP losses.

Own counterpart of the reference's loop -- `Worker.train_epoch` (torchext/worker.py:362-443) with the loss of
`exp_synph.Worker.loss_forward` (model/exp_synph.py:93-118): per step

    copy_data -> net_forward -> loss_forward -> backward -> optimizer.step

with the reference's bucket names for the timings.  The loss is the photometric pattern similarity
(`RectifiedPatternSimilarityLoss`, block 9, census_sad, eps 0.5, masked by the LCN std) plus
`dp_weight` (0.02, exp_synph.py:41) times the edge-aware disparity loss with `1 - sigmoid(edge)`.
The reference's CNN (`DispEdgeDecoders`) is stock PyTorch and stays stock PyTorch on ROCm; the small
network below only stands in for it where no checkpoint or dataset is available (offline box).
"""
import time

import torch

from . import torchext as te
from . import sharding


class StopWatch:
    """Accumulating named timers (reference: torchext/worker.py StopWatch), synchronising the device per bucket
    exactly where the reference does (worker.py:391,404,410,415)."""

    def __init__(self, device=None):
        self.device = device
        self.t0, self.total, self.count = {}, {}, {}

    def start(self, name):
        self.t0[name] = time.perf_counter()

    def stop(self, name, sync=True):
        if sync and self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - self.t0.pop(name)
        self.total[name] = self.total.get(name, 0.0) + dt
        self.count[name] = self.count.get(name, 0) + 1

    def mean_ms(self):
        return {k: 1e3 * v / self.count[k] for k, v in self.total.items()}


class SmallDispEdgeNet(torch.nn.Module):
    """IR image -> (disparity in [0, max_disp], edge logits); a few 3x3 convs (MIOpen), CoordConv first like
    the reference's decoders feed on (networks.py:179)."""

    def __init__(self, max_disp=64, width=16):
        super().__init__()
        self.max_disp = max_disp
        self.body = torch.nn.Sequential(
            te.CoordConv2d(1, width, 3, 1, 1), torch.nn.ReLU(),
            torch.nn.Conv2d(width, width, 3, 1, 1), torch.nn.ReLU(),
            torch.nn.Conv2d(width, width, 3, 1, 2, dilation=2), torch.nn.ReLU())
        self.disp_head = torch.nn.Conv2d(width, 1, 3, 1, 1)
        self.edge_head = torch.nn.Conv2d(width, 1, 3, 1, 1)

    def forward(self, x):
        f = self.body(x)
        return torch.sigmoid(self.disp_head(f)) * self.max_disp, self.edge_head(f)


class DisparityTrainer:
    """One data-parallel replica.  `pattern` [1,C,H,W] is the LCN'd reference pattern (exp_synph.py:64-71)."""

    def __init__(self, net, pattern, im_height, im_width, lr=1e-3, dp_weight=0.02, lcn_radius=5, lcn_eps=0.05,
                 algo="fast", process_group=None):
        self.device = pattern.device
        self.net = net.to(self.device)
        self.lcn = te.LCN(lcn_radius, lcn_eps)
        self.photo = te.RectifiedPatternSimilarityLoss(im_height, im_width, pattern, loss_type="census_sad", loss_eps=0.5,
                                                       algo=algo)
        self.disparity_loss = te.DisparityLoss()
        self.dp_weight = dp_weight
        self.optimizer = torch.optim.Adam(self.net.parameters(), lr=lr)
        self.pg = process_group
        self.watch = StopWatch(self.device)

    def loss_forward(self, disp, edge, im_lcn, std):
        """exp_synph.py:106-118 (single scale): photometric term + dp_weight * disparity term"""
        if self.pg is not None:
            num, den, _ = self.photo.terms(disp, im_lcn, std)
            photo = sharding.reduce_ratio(num, den, self.pg)          # ratio of all-reduced sums (networks.py:377)
        else:
            photo, _ = self.photo(disp, im_lcn, std)
        vals = [photo]
        if self.dp_weight > 0:
            edge0 = 1 - torch.sigmoid(edge)
            vals.append(self.disparity_loss(disp, edge0) * self.dp_weight)
        return vals

    def train_step(self, ir):
        """ir [B,1,H,W] raw IR frames already on the device (copy_data is the caller's H2D)."""
        w = self.watch
        w.start("total")
        self.optimizer.zero_grad(set_to_none=True)
        w.start("forward")
        with torch.no_grad():
            im_lcn, std = self.lcn(ir)
        disp, edge = self.net(im_lcn)
        w.stop("forward")
        w.start("loss")
        vals = self.loss_forward(disp, edge, im_lcn, std)
        err = sum(vals)
        w.stop("loss")
        w.start("backward")
        err.backward()
        if self.pg is not None:
            for p in self.net.parameters():                            # plain DDP-style gradient averaging
                if p.grad is not None:
                    torch.distributed.all_reduce(p.grad, group=self.pg)
                    p.grad /= torch.distributed.get_world_size(self.pg)
        w.stop("backward")
        w.start("optimizer")
        self.optimizer.step()
        w.stop("optimizer")
        w.stop("total", sync=False)
        return [float(v.detach()) for v in vals]
