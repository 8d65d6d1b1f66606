"""Pose-sequence VAE (core/networks/poses_reconstruction/autoencoder.py:8-92) on the gfx950 kernels."""
import torch
from torch import nn

from .... import ops
from ..building_blocks import ConvNormRelu, conv_head, make_head

_ENC_DOWN = (False, False, True, True, True, True, True)  # autoencoder.py:17-25


class PoseSeqEncoder(nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        a = cfg.POSE2POSE.AUTOENCODER
        cin = cfg.DATASET.NUM_LANDMARKS * 2
        chans = [cin] + [256] * 6 + [a.CODE_DIM * 2]
        self.blocks = nn.Sequential(*[
            ConvNormRelu('1d', chans[i], chans[i + 1], downsample=_ENC_DOWN[i], norm=a.NORM, leaky=a.LEAKY_RELU)
            for i in range(7)])

    def forward(self, x):
        """(B,T,2,K) -> mu (B,D), logvar (B,D): even / odd channels of the first remaining time step."""
        h = x.reshape(x.shape[0], x.shape[1], -1)
        for block in self.blocks:
            h = block.forward_cl(h)
        h = h[:, 0, :]  # F.interpolate(x, 1) (nearest) keeps time step 0, autoencoder.py:31
        return h[:, 0::2], h[:, 1::2]


    def forward_pair(self, xa, xb):
        """forward(xa), forward(xb) -- the two no-grad calls of a train step (voice2pose.py:160-176) -- as ONE pass over the concatenated batch:
        half the launches of a stack whose launches are latency-bound.  BatchNorm blocks in training mode normalise each half with its own batch
        statistics and update their running statistics half by half (``bn_groups=2``), so results and buffers equal the two calls'."""
        assert xa.shape == xb.shape and not (torch.is_grad_enabled() and (xa.requires_grad or xb.requires_grad))
        B = xa.shape[0]
        h = torch.cat([xa.reshape(B, xa.shape[1], -1), xb.reshape(B, xb.shape[1], -1)], 0)
        for block in self.blocks:
            h = block.forward_cl(h, bn_groups=2)
        h = h[:, 0, :]
        return (h[:B, 0::2], h[:B, 1::2]), (h[B:, 0::2], h[B:, 1::2])


class PoseSeqDecoder(nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        a = cfg.POSE2POSE.AUTOENCODER
        for j, i in enumerate((5, 4, 3, 2, 1)):
            setattr(self, 'd%d' % i, ConvNormRelu('1d', a.CODE_DIM if j == 0 else 256, 256, downsample=False, norm=a.NORM, leaky=a.LEAKY_RELU))
        self.blocks = nn.Sequential(
            *[ConvNormRelu('1d', 256, 256, downsample=False, norm=a.NORM, leaky=a.LEAKY_RELU) for _ in range(4)],
            make_head(256, cfg.DATASET.NUM_LANDMARKS * 2, 1))

    def forward_cl(self, code):
        """(B,D) -> (B,64,2K) channels-last: nearest x2, then 5 x (linear x2 -> block), 4 blocks, k1 head."""
        h = code.unsqueeze(1).expand(-1, 2, -1).contiguous()  # F.interpolate(x.unsqueeze(-1), 2), autoencoder.py:60
        for i in (5, 4, 3, 2, 1):
            h = getattr(self, 'd%d' % i).forward_cl(ops.UpsampleAddFn.apply(h, None, h.shape[1] * 2))
        for block in list(self.blocks)[:4]:
            h = block.forward_cl(h)
        return conv_head(h, self.blocks[4])

    def forward(self, x):
        return ops.cf_view(self.forward_cl(x))


class Autoencoder(nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        self.cfg = cfg
        self.encoder = PoseSeqEncoder(cfg)
        self.decoder = PoseSeqDecoder(cfg)

    def forward(self, x, num_frames, mel=None, external_code=None):
        K = self.cfg.DATASET.NUM_LANDMARKS
        if external_code is not None:
            out = self.decoder.forward_cl(external_code).reshape(-1, num_frames, 2, K)
            return out, external_code, torch.zeros_like(external_code)
        mu, logvar = self.encoder(x)
        eps = torch.randn(logvar.shape, device=logvar.device)
        code = mu + torch.exp(0.5 * logvar) * eps  # reparameterisation, autoencoder.py:86-87
        out = self.decoder.forward_cl(code).reshape(-1, num_frames, 2, K)
        return out, mu.squeeze(-1), logvar.squeeze(-1)
