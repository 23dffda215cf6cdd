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

    def __init__(self, radius, epsilon):
        super().__init__()
        self.radius = radius
        self.epsilon = epsilon

    def forward(self, data):
        return lcn(data.contiguous(), self.radius, self.epsilon)  # noqa: F405
