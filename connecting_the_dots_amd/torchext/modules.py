"""Mirror of the reference's `torchext/modules.py` (CoordConv2d, modules.py:7-27)."""
import torch

from .functions import *  # noqa: F401,F403  (the reference re-exports functions here, modules.py:5)


class CoordConv2d(torch.nn.Module):
    """Conv2d over the input concatenated with a normalised (u, v) coordinate grid in [-1, 1]."""

    def __init__(self, channels_in, channels_out, kernel_size, stride, padding):
        super().__init__()
        self.conv = torch.nn.Conv2d(channels_in + 2, channels_out, kernel_size=kernel_size, padding=padding,
                                    stride=stride)
        self.uv = None

    def forward(self, x):
        height, width = x.shape[2], x.shape[3]
        if self.uv is None or self.uv.shape[-2:] != (height, width):
            # float64 linspace then cast, as the reference builds the grid in numpy float64
            u = 2 * torch.arange(width, dtype=torch.float64) / (width - 1) - 1
            v = 2 * torch.arange(height, dtype=torch.float64) / (height - 1) - 1
            uv = torch.stack((u.view(1, -1).expand(height, -1), v.view(-1, 1).expand(-1, width)))
            self.uv = uv.reshape(1, 2, height, width).to(torch.float32)
        self.uv = self.uv.to(x.device)
        uv = self.uv.expand(x.shape[0], *self.uv.shape[1:])
        return self.conv(torch.cat((x, uv), dim=1))


class LCN(torch.nn.Module):
    """Local contrast normalisation with the constructor / call signature of the reference's
    `networks.LCN(radius, epsilon)` (model/networks.py:507-533); one fused HIP kernel."""

    def __init__(self, radius, epsilon, algo=None):
        super().__init__()
        self.radius = radius
        self.epsilon = epsilon
        self.algo = algo                  # additive: None / 'exact' (f64 box sums) | 'fast' (f32 sliding sums, radius 5)

    def forward(self, data):
        return lcn(data.contiguous(), self.radius, self.epsilon, self.algo)  # noqa: F405


class RectifiedPatternSimilarityLoss(torch.nn.Module):
    """Photometric loss of the reference (`networks.RectifiedPatternSimilarityLoss`, model/networks.py:340-378),
    same constructor and call signature: warp the (channel-averaged) reference pattern by the predicted
    disparity along u, compare with the image through the block photometric loss (block 9), and return
    `((mask * diff).sum() / mask.sum(), pattern_proj)` with `mask = std` (or ones).

    The warp is `grid_sample(bilinear, border)` exactly as the reference calls it today (normalised with W-1 /
    H-1 but sampled with align_corners=False -- SURVEY 7.3-6); the block loss and its gradient are the HIP
    kernels of photometric.hip.  Under data parallelism reduce numerator and denominator separately
    (`connecting_the_dots_amd.sharding.reduce_ratio`)."""

    def __init__(self, im_height, im_width, pattern, loss_type='census_sad', loss_eps=0.5, algo=None):
        super().__init__()
        self.algo = algo                      # additive: 'fast' (default) | 'exact', see photometric_loss
        self.im_height = im_height
        self.im_width = im_width
        self.pattern = pattern.mean(dim=1, keepdim=True).contiguous()
        u = torch.arange(im_width, dtype=torch.float32).view(1, 1, -1).expand(1, im_height, -1)
        v = torch.arange(im_height, dtype=torch.float32).view(1, -1, 1).expand(1, -1, im_width)
        self.u0, self.v0 = u.contiguous(), v.contiguous()
        self.loss_type = loss_type
        self.loss_eps = loss_eps

    def terms(self, disp0, im, std=None):
        """(numerator, denominator, pattern_proj) of the masked mean, for cross-rank reduction."""
        dev = disp0.device
        self.pattern, self.u0, self.v0 = self.pattern.to(dev), self.u0.to(dev), self.v0.to(dev)
        B = disp0.shape[0]
        if self._fused(disp0, im, std):
            # algo='fast': warp + block loss + masked sums in one kernel each way (photometric_fast.hip)
            _, pattern_proj, terms = pattern_loss(disp0.contiguous(), im.contiguous(),  # noqa: F405
                                                  None if std is None else std.contiguous(), self.pattern,
                                                  self.loss_type, self.loss_eps)
            return terms[0], terms[1], pattern_proj
        u1 = self.u0 - disp0.contiguous().view(B, self.im_height, self.im_width)
        gx = 2 * (u1 / (self.im_width - 1) - 0.5)
        gy = (2 * (self.v0 / (self.im_height - 1) - 0.5)).expand(B, -1, -1)
        grid = torch.stack((gx, gy), dim=3)
        pattern = self.pattern.expand(B, *self.pattern.shape[1:])
        pattern_proj = torch.nn.functional.grid_sample(pattern, grid, padding_mode='border', align_corners=False)
        mask = torch.ones_like(im)
        if std is not None:
            mask = mask * std
        diff = photometric_loss(pattern_proj.contiguous(), im.contiguous(), 9, self.loss_type, self.loss_eps,  # noqa: F405
                                algo=self.algo)
        return (mask * diff).sum(), mask.sum(), pattern_proj

    def _fused(self, disp0, im, std):
        import os
        algo = self.algo or os.environ.get("CTD_PHOTO_ALGO", "fast")
        return (algo == "fast" and disp0.dtype == torch.float32 and disp0.dim() == 4 and disp0.shape[1] == 1
                and im.shape == disp0.shape and (std is None or std.shape == disp0.shape)
                and not im.requires_grad and (std is None or not std.requires_grad))

    def forward(self, disp0, im, std=None):
        if self._fused(disp0, im, std):
            self.pattern = self.pattern.to(disp0.device)
            val, pattern_proj, _ = pattern_loss(disp0.contiguous(), im.contiguous(),  # noqa: F405
                                                None if std is None else std.contiguous(), self.pattern,
                                                self.loss_type, self.loss_eps)
            return val, pattern_proj
        num, den, pattern_proj = self.terms(disp0, im, std)
        return num / den, pattern_proj


class MultiScalePatternSimilarityLoss(torch.nn.Module):
    """Additive: the `RectifiedPatternSimilarityLoss` of every pyramid level (the loop of
    model/exp_synph.py:107-111) in one forward and one backward launch.  `patterns[s]` is the LCN'd pattern of
    scale s, [1,C,H_s,W_s]; call with lists `(disps, ims, stds)` (stds entries may be None) ->
    `(vals [n_scales], [pattern_proj per scale])`.  Values equal the per-scale fused module's bit for bit."""

    def __init__(self, patterns, loss_type='census_sad', loss_eps=0.5):
        super().__init__()
        self.patterns = [p.mean(dim=1, keepdim=True).contiguous() for p in patterns]
        self.loss_type, self.loss_eps = loss_type, loss_eps

    def forward(self, disps, ims, stds=None):
        n = len(self.patterns)
        stds = list(stds) if stds is not None else [None] * n
        self.patterns = [p.to(disps[0].device) for p in self.patterns]
        vals, _, projs = pattern_loss_multi(list(disps), list(ims), stds, self.patterns, self.loss_type,  # noqa: F405
                                            self.loss_eps)
        return vals, projs

    def terms(self, disps, ims, stds=None):
        """(terms [n_scales, 3] = (numerator, denominator, ratio) per scale, [pattern_proj per scale]): numerator and
        denominator of every scale's masked mean, for the cross-rank ratio of sums (sharding.reduce_ratio_ddp)."""
        n = len(self.patterns)
        stds = list(stds) if stds is not None else [None] * n
        self.patterns = [p.to(disps[0].device) for p in self.patterns]
        _, terms, projs = pattern_loss_multi(list(disps), list(ims), stds, self.patterns, self.loss_type,  # noqa: F405
                                             self.loss_eps)
        return terms, projs


class DispToDepth(torch.nn.Module):
    """`networks.DispToDepth(focal_length, baseline)` (model/networks.py:313-321), one fused kernel each way."""

    def __init__(self, focal_length, baseline):
        super().__init__()
        self.baseline_focal_length = baseline * focal_length

    def forward(self, disp):
        return disp_to_depth(disp, self.baseline_focal_length)  # noqa: F405


class DisparityLoss(torch.nn.Module):
    """`networks.DisparityLoss()` (model/networks.py:380-412): Sobel + Laplace-mixture NLL (with `edge`) or
    mean clamped gradient magnitude (without), fused forward and backward."""

    def forward(self, disp, edge=None):
        return disparity_loss(disp, edge)  # noqa: F405


class ProjectionDepthSimilarityLoss(torch.nn.Module):
    """`networks.ProjectionDepthSimilarityLoss(K, Ki, im_height, im_width, clamp=-1)` (model/networks.py:416-503),
    same constructor and call signature `(depth0, depth1, R0, t0, R1, t1) -> scalar`."""

    def __init__(self, K, Ki, im_height, im_width, clamp=-1):
        super().__init__()
        self.K = K.reshape(3, 3).to(torch.float32).contiguous()
        self.im_height, self.im_width, self.clamp = im_height, im_width, clamp
        # rays in float64 then float32, as the reference builds them with numpy (networks.py:428-434)
        u = torch.arange(im_width, dtype=torch.float64).view(1, -1).expand(im_height, -1)
        v = torch.arange(im_height, dtype=torch.float64).view(-1, 1).expand(-1, im_width)
        uv1 = torch.stack((u, v, torch.ones_like(u)), dim=2).reshape(-1, 3)
        self.ray = (uv1 @ Ki.reshape(3, 3).to(torch.float64).cpu().T).to(torch.float32).contiguous()

    def forward(self, depth0, depth1, R0, t0, R1, t1):
        dev = depth0.device
        self.K, self.ray = self.K.to(dev), self.ray.to(dev)
        return geometric_loss(depth0, depth1, self.ray, self.K, R0, t0, R1, t1, self.clamp)  # noqa: F405
