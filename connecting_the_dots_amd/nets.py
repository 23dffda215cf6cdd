"""Disparity + edge network of BASELINE config 5, written from the reference's layer list.

Output contract of `networks.DispEdgeDecoders(channels_in=2, max_disp=128, imsizes, output_ms=True)`
(model/networks.py:122-310, train_val.py:19-22): for an input [N, 2, H, W] (LCN'd IR frame and raw frame)

    ([disp_0 .. disp_3], [edge_0 .. edge_2])

with disp_s [N, 1, H/2^s, W/2^s] = sigmoid(conv(x) - 3) * max_disp / 2^s (`SigmoidAffine`, networks.py:90-99 with the
factory parameters of :296) and edge_s [N, 1, H/2^s, W/2^s] plain logits.  The disparity decoder is a 7-level
hourglass (encoder widths 32..512, kernel 7 / 5 / 3, two convs per level, the first with stride 2; decoder with
stride-2 transposed convs, skip connections, and from the third prediction on the upsampled coarser disparity as an
extra input channel); the edge decoder is the same hourglass cut at 3 levels.  The CNN itself is stock PyTorch
(MIOpen convolutions) -- SURVEY 8e: "standard DDP, not part of the hand-written path" -- what this file owns is the
architecture and its output contract.
"""
import torch
import torch.nn.functional as F

ENC_WIDTHS = (32, 64, 128, 256, 512, 512, 512)
ENC_KERNELS = (7, 5, 3, 3, 3, 3, 3)
DEC_WIDTHS = (512, 512, 256, 128, 64, 32, 16)      # decoder level 7 (coarsest) .. 1 (full resolution)


def _down(cin, cout, k):
    p = (k - 1) // 2
    return torch.nn.Sequential(torch.nn.Conv2d(cin, cout, k, stride=2, padding=p), torch.nn.ReLU(inplace=True),
                               torch.nn.Conv2d(cout, cout, k, padding=p), torch.nn.ReLU(inplace=True))


def _up(cin, cout):
    return torch.nn.Sequential(torch.nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1),
                               torch.nn.ReLU(inplace=True))


def _fuse(cin, cout):
    return torch.nn.Sequential(torch.nn.Conv2d(cin, cout, 3, padding=1), torch.nn.ReLU(inplace=True))


def _crop(x, ref):
    return x[:, :, :ref.shape[2], :ref.shape[3]]


class Hourglass(torch.nn.Module):
    """`depth` encoder levels (of the 7 of ENC_WIDTHS) and as many decoder levels; predictions at the `n_out` finest
    decoder levels, each fed back (bilinearly upsampled) into the next finer level.  `head(s, channels)` builds the
    prediction layer of scale s."""

    def __init__(self, channels_in, depth, n_out, head):
        super().__init__()
        assert 1 <= n_out <= depth <= len(ENC_WIDTHS)
        self.depth, self.n_out = depth, n_out
        enc_in = (channels_in,) + ENC_WIDTHS[:depth - 1]
        self.enc = torch.nn.ModuleList(_down(enc_in[l], ENC_WIDTHS[l], ENC_KERNELS[l]) for l in range(depth))
        dec_w = DEC_WIDTHS[len(DEC_WIDTHS) - depth:]            # widths of decoder levels depth .. 1
        self.up, self.fuse, self.heads = torch.nn.ModuleList(), torch.nn.ModuleList(), torch.nn.ModuleDict()
        for k in range(depth):                                  # k = 0: coarsest decoder level (= level `depth`)
            level = depth - k                                   # output resolution H / 2^(level-1)
            cin = ENC_WIDTHS[depth - 1] if k == 0 else dec_w[k - 1]
            self.up.append(_up(cin, dec_w[k]))
            skip = ENC_WIDTHS[level - 2] if level >= 2 else 0   # encoder feature of the same resolution
            fed_back = 1 if level < n_out else 0                # upsampled prediction of the coarser scale
            self.fuse.append(_fuse(dec_w[k] + skip + fed_back, dec_w[k]))
            if level <= n_out:
                self.heads[str(level - 1)] = head(level - 1, dec_w[k])

    def forward(self, x):
        feats = []
        h = x
        for e in self.enc:
            h = e(h)
            feats.append(h)
        outs, prev = {}, None
        for k in range(self.depth):
            level = self.depth - k
            ref = feats[level - 2] if level >= 2 else x
            parts = [_crop(self.up[k](h), ref)]
            if level >= 2:
                parts.append(ref)
            if prev is not None:
                parts.append(_crop(F.interpolate(prev, scale_factor=2, mode="bilinear", align_corners=False), ref))
            h = self.fuse[k](torch.cat(parts, 1))
            if str(level - 1) in self.heads:
                prev = self.heads[str(level - 1)](h)
                outs[level - 1] = prev
        return [outs[s] for s in range(self.n_out)]


class _DispHead(torch.nn.Module):
    def __init__(self, channels, alpha):
        super().__init__()
        self.conv = torch.nn.Conv2d(channels, 1, 3, padding=1)
        self.alpha = alpha

    def forward(self, x):
        return torch.sigmoid(self.conv(x) - 3.0) * self.alpha      # SigmoidAffine(alpha, beta 0, gamma 1, offset 3)


class DispEdgeNet(torch.nn.Module):
    """Disparity hourglass (7 levels, 4 scales out) + edge hourglass (3 levels, 3 scales out)."""

    def __init__(self, channels_in=2, max_disp=128):
        super().__init__()
        self.disp = Hourglass(channels_in, 7, 4, lambda s, c: _DispHead(c, max_disp / 2 ** s))
        self.edge = Hourglass(channels_in, 3, 3, lambda s, c: torch.nn.Conv2d(c, 1, 3, padding=1))

    def forward(self, x):
        return self.disp(x), self.edge(x)
