"""Motion patch discriminator (core/networks/keypoints_generation/discriminator.py:6-23) on the gfx950 kernels."""
from torch import nn

from ..building_blocks import ConvNormRelu, conv_head, make_head


class PoseSequenceDiscriminator(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        leaky = self.cfg.VOICE2POSE.POSE_DISCRIMINATOR.LEAKY_RELU
        self.seq = nn.Sequential(
            ConvNormRelu('1d', cfg.DATASET.NUM_LANDMARKS * 2, 256, downsample=True, leaky=leaky),
            ConvNormRelu('1d', 256, 512, downsample=True, leaky=leaky),
            ConvNormRelu('1d', 512, 1024, kernel_size=3, stride=1, padding=1, leaky=leaky),
            make_head(1024, 1, 3, 1, 1))

    def forward(self, x):
        """(B,T',2,K) -> patch scores (B,T'')."""
        h = x.reshape(x.size(0), x.size(1), -1)  # (B,T',2K) is already channels-last
        for block in list(self.seq)[:3]:
            h = block.forward_cl(h)
        return conv_head(h, self.seq[3]).squeeze(-1)
