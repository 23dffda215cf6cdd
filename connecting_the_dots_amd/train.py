"""Training steps of the disparity network on the HIP losses: `DisparityTrainer` (one
scale, small network: the plumbing test) and `TrackTrainer` (BASELINE config 5 at its real shape: four scales,
photometric + disparity + edge + geometric terms on frame tracks, the full-size network, data parallel).

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
            photo = sharding.reduce_ratio_ddp(num, den, self.pg)      # ratio of all-reduced sums (networks.py:377)
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


class TrackTrainer:
    """BASELINE config 5: one replica of the reference's multi-scale photometric + geometric training step
    (`exp_synphge.Worker`, model/exp_synphge.py:133-202, on the loop of torchext/worker.py:362-443).

    A batch is a dict of device tensors shaped like the reference's after `copy_data` (exp_synphge.py:93-115):
        im{s}   [tl, B, 1, H_s, W_s]  raw IR frames of scale s = 0..3 (H_s = H / 2^s)
        grad{s} [tl, B, 1, H_s, W_s]  gradient magnitude of the ground-truth disparity, scales 0..2 (edge supervision)
        id      [B]                   sample ids; the edge decoder is supervised where id > train_edge
        R [tl, B, 3, 3], t [tl, B, 3] camera poses of the track frames
    Terms, in the reference's order (exp_synphge.py:141-200): photometric pattern similarity of every scale (all four
    in one fused launch each way), `dp_weight` * edge-aware disparity loss at scale 0, BCE-with-logits edge loss of
    scales 0..2 (pos_weight 0.1) on the supervised samples, and for every scale and every frame pair of the track the
    two-view geometric loss on disparity -> depth, weighted `ge_weight / (tl (tl-1) / 2)`.

    Data parallel: wrap happens here (`torch.nn.parallel.DistributedDataParallel`, bucketed gradient all-reduce
    overlapped with backward, RCCL).  DDP averages the ranks' gradients; the terms that are plain means over equal
    shards (disparity loss, geometric loss) need nothing more, the two that are RATIOS OF SUMS over the batch -- the
    mask-weighted photometric mean (networks.py:377) and the edge loss over the supervised samples only
    (exp_synph.py:120-131), whose count differs from rank to rank -- reduce numerator and denominator across the ranks
    (`sharding.reduce_ratio_ddp`: value = the reference's batch value on every rank, averaged gradient = the reference's
    batch gradient).  `tests/test_config5_ddp_gpu.py` holds a two-rank step against the one-process step on the joined batch.
    """

    def __init__(self, net, patterns, K, baseline, focal_lengths, dp_weight=0.02, ge_weight=0.1, train_edge=-1, lr=1e-4,
                 lcn_radius=5, lcn_eps=0.05, process_group=None, device_ids=None):
        self.device = patterns[0].device
        self.imsizes = [(p.shape[-2], p.shape[-1]) for p in patterns]
        self.net = net.to(self.device)
        self.pg = process_group
        self.model = self.net
        if process_group is not None:
            self.model = torch.nn.parallel.DistributedDataParallel(self.net, device_ids=device_ids,
                                                                   process_group=process_group)
        self.lcn = te.LCN(lcn_radius, lcn_eps)
        self.photo = te.MultiScalePatternSimilarityLoss(patterns, loss_type="census_sad", loss_eps=0.5)
        self.disparity_loss = te.DisparityLoss()
        self.edge_loss = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([0.1], device=self.device))
        self.d2d, self.geo = [], []
        for s, (h, w) in enumerate(self.imsizes):
            Ks = K.clone().double()
            Ks[:2] = Ks[:2] / 2 ** s                                  # intrinsics of scale s
            self.geo.append(te.ProjectionDepthSimilarityLoss(Ks.float(), torch.linalg.inv(Ks).float(), h, w, clamp=0.1))
            self.d2d.append(te.DispToDepth(float(focal_lengths[s]), float(baseline)))
        self.dp_weight, self.ge_weight, self.train_edge = dp_weight, ge_weight, train_edge
        self.optimizer = torch.optim.Adam(self.net.parameters(), lr=lr)
        self.watch = StopWatch(self.device)

    def copy_data(self, batch):
        """LCN of every scale's frames (exp_synphge.py:107-115): returns per scale the network-side tensors"""
        data = dict(batch)
        with torch.no_grad():
            for s in range(len(self.imsizes)):
                im = batch["im%d" % s]
                tl, B = im.shape[:2]
                lcn, std = self.lcn(im.reshape(tl * B, *im.shape[2:]).contiguous())
                data["lcn%d" % s], data["std%d" % s] = lcn, std
        return data

    def net_forward(self, data):
        im = data["im0"]
        x = torch.cat((data["lcn0"], im.reshape(-1, *im.shape[2:])), dim=1)     # [tl*B, 2, H, W]
        return self.model(x)

    def loss_forward(self, out, data, train=True):
        disps, edges = out
        tl, B = data["im0"].shape[:2]
        n = len(self.imsizes)
        lcns, stds = [data["lcn%d" % s] for s in range(n)], [data["std%d" % s] for s in range(n)]
        if self.pg is not None:
            terms, _ = self.photo.terms(disps, lcns, stds)                        # [n, 3]: numerator, denominator, ratio
            vals = [sharding.reduce_ratio_ddp(terms[s, 0], terms[s, 1], self.pg) for s in range(n)]
        else:
            vals = list(self.photo(disps, lcns, stds)[0])
        if self.dp_weight > 0:
            vals.append(self.disparity_loss(disps[0], 1 - torch.sigmoid(edges[0])) * self.dp_weight)
        # Edge loss on the supervised samples (exp_synph.py:120-131: `edge_loss(e[mask], grad[mask])`, zero when the
        # mask is empty) as a mask-weighted mean: no host round trip to look at the mask, and a rank without a supervised
        # sample still sends (zero) gradients to every edge-decoder parameter, which DDP requires of every step.
        sup = (data["id"] > self.train_edge).to(torch.float32).view(1, B, 1, 1, 1)
        for s, e in enumerate(edges):
            e5 = e.view(tl, B, *e.shape[1:])
            gt = (data["grad%d" % s] < 0.2).to(torch.float32)                     # inverse edge map: 0 = edge
            bce = torch.nn.functional.binary_cross_entropy_with_logits(e5, gt, pos_weight=self.edge_loss.pos_weight,
                                                                       reduction="none")
            num = (bce * sup).sum()
            cnt = sup.sum() * float(tl * e5[0, 0].numel())
            if self.pg is not None:
                vals.append(sharding.reduce_ratio_ddp(num, cnt, self.pg))
            else:
                vals.append(num / cnt.clamp_min(1.0))
        if not train:
            return vals
        ge_num = tl * (tl - 1) / 2
        R, t = data["R"], data["t"]
        for s in range(n):
            depth = self.d2d[s](disps[s]).view(tl, B, *disps[s].shape[1:])
            for i0 in range(tl):
                for i1 in range(i0 + 1, tl):
                    v = self.geo[s](depth[i0].contiguous(), depth[i1].contiguous(), R[i0].contiguous(), t[i0].contiguous(),
                                    R[i1].contiguous(), t[i1].contiguous())
                    vals.append(v * (self.ge_weight / ge_num))
        return vals

    @torch.no_grad()
    def evaluate(self, batch, metric=None):
        """The test pass of the reference (`Worker.test_epoch`, torchext/worker.py:464-500, with the callbacks of
        exp_synph.py:201-224): forward without gradients, the loss terms of `loss_forward(train=False)` (no geometric
        terms), and the disparity error of scale 0 against `batch["disp0"]` [tl, B, 1, H, W] on the evaluation crop.
        Returns (loss values, metric dict); pass a `DisparityMetric` to accumulate over several batches.  With a
        process group the loss terms reduce across the ranks (as in training): every rank has to make the call; the
        metric is that of the rank's own shard."""
        was_training = self.net.training
        self.net.eval()
        try:
            data = self.copy_data(batch)
            out = self.net(torch.cat((data["lcn0"], data["im0"].reshape(-1, *data["im0"].shape[2:])), dim=1))
            vals = self.loss_forward(out, data, train=False)
            metric = metric if metric is not None else DisparityMetric()
            es = out[0][0]
            metric.add(es, batch["disp0"].reshape(es.shape).to(es.dtype))
        finally:
            self.net.train(was_training)
        return [float(v) for v in vals], metric.get()

    def train_step(self, batch):
        w = self.watch
        w.start("total")
        self.optimizer.zero_grad(set_to_none=True)
        w.start("data")
        data = self.copy_data(batch)
        w.stop("data")
        w.start("forward")
        out = self.net_forward(data)
        w.stop("forward")
        w.start("loss")
        vals = self.loss_forward(out, data)
        err = sum(vals)
        w.stop("loss")
        w.start("backward")
        err.backward()                                   # DDP all-reduces the gradient buckets under this
        w.stop("backward")
        w.start("optimizer")
        self.optimizer.step()
        w.stop("optimizer")
        w.stop("total", sync=False)
        return [float(v.detach()) for v in vals]


class DisparityMetric:
    """Disparity error of the evaluation pass: the reference's `co.metric.MultipleMetric(DistanceMetric(vec_length=1),
    OutlierFractionMetric(vec_length=1, thresholds=[0.1, 0.5, 1, 2, 5]))` (model/exp_synph.py:201-205,
    co/metric.py:76-129) on the region where the reference evaluates, rows 13 .. H-13 and columns 140 .. W-13
    (exp_synph.py:46-50, `crop_output` :226-232), over the pixels with ground truth (`gt > 0`, exp_synph.py:147).
    Same keys as `metric.get()` there: dist2_mean / _std / _median / _q10 / _q90 / _min / _max and of<threshold>.
    Distances are gathered on the device; `get()` evaluates them once."""

    THRESHOLDS = (0.1, 0.5, 1, 2, 5)

    def __init__(self, crop=(13, 13, 140, 13)):
        self.crop = crop                      # rows cut at the top / bottom, columns cut at the left / right
        self.dists = []

    def reset(self):
        self.dists = []

    def add(self, es, gt):
        """es, gt [..., H, W] estimated and ground-truth disparity of scale 0"""
        t, b, l, r = self.crop
        H, W = es.shape[-2:]
        es, gt = es[..., t:H - b, l:W - r], gt[..., t:H - b, l:W - r]
        self.dists.append((es - gt).abs()[gt > 0].reshape(-1).to(torch.float32))

    def get(self):
        d = torch.cat(self.dists).double()
        if d.numel() == 0:
            return {}
        q = torch.quantile(d, torch.tensor([0.1, 0.5, 0.9], dtype=d.dtype, device=d.device)) if d.numel() <= (1 << 24) \
            else torch.tensor([float("nan")] * 3)
        out = {"dist2_mean": float(d.mean()), "dist2_std": float(d.std(unbiased=False)), "dist2_median": float(q[1]),
               "dist2_q10": float(q[0]), "dist2_q90": float(q[2]), "dist2_min": float(d.min()), "dist2_max": float(d.max())}
        for t in self.THRESHOLDS:
            out["of%s" % t] = float((d > t).double().mean())
        return out
