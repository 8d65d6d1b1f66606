"""ConvNormRelu over the gfx950 kernels -- same constructor, state_dict keys and forward contract as the
reference block (core/networks/building_blocks.py:4-55): Conv{1,2}d(bias=False) -> BatchNorm | InstanceNorm
-> LeakyReLU(0.2) | ReLU, Kaiming-normal weights.

``self.conv`` / ``self.norm`` are torch modules used purely as PARAMETER CONTAINERS (so checkpoints written
by the reference load with strict=True); their forward() is never called.  The convolution weight keeps its
logical (Cout,Cin,*k) shape but lives in (Cout,*k,Cin) memory, the layout the implicit-GEMM kernels read.
Internally tensors are channels-last; ``forward`` accepts/returns the reference's channels-first logical
shapes as zero-copy views, ``forward_cl`` is the channels-last fast path used between blocks.
"""
import torch
from torch import nn

from ... import ops


class ConvNormRelu(nn.Module):
    def __init__(self, conv_type='1d', in_channels=3, out_channels=64, downsample=False,
                 kernel_size=None, stride=None, padding=None, norm='BN', leaky=False):
        super().__init__()
        if kernel_size is None:  # the two stock shapes, building_blocks.py:8-12
            kernel_size, stride, padding = (4, 2, 1) if downsample else (3, 1, 1)
        if conv_type == '2d':
            conv_cls, bn_cls, in_cls = nn.Conv2d, nn.BatchNorm2d, nn.InstanceNorm2d
        elif conv_type == '1d':
            conv_cls, bn_cls, in_cls = nn.Conv1d, nn.BatchNorm1d, nn.InstanceNorm1d
        else:
            raise NotImplementedError(conv_type)
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        if norm == 'BN':
            self.norm = bn_cls(out_channels)
        elif norm == 'IN':
            self.norm = in_cls(out_channels)
        else:
            raise NotImplementedError
        nn.init.kaiming_normal_(self.conv.weight)
        self.conv.weight.data = ops.to_weight_layout(self.conv.weight.data)
        self.conv_type, self.norm_type = conv_type, norm
        self.stride, self.padding = int(stride), int(padding)
        self.slope = ops.LEAKY_SLOPE if leaky else 0.0

    def _is_l0_block(self):
        c = self.conv
        return (self.conv_type == '2d' and c.in_channels == 1 and c.out_channels == 64 and tuple(c.kernel_size) == (3, 3)
                and self.stride == 1 and self.padding == 1 and (self.norm_type == 'IN' or self.training))

    def forward_cl(self, x_cl, in_holder=None, out_holder=None, out_f32=False, bn_groups=1):
        """(B,H,W,Cin)|(B,T,Cin) channels-last -> (B,Ho,Wo,Cout)|(B,To,Cout) channels-last.
        ``in_holder`` / ``out_holder`` (ops.NormBwdHolder, 2-D blocks of a strictly sequential chain only): ``x_cl`` is the output of
        the normalisation that filled ``in_holder`` and has no other consumer; this block's normalisation fills ``out_holder``.
        bf16-storage path (ops.STORAGE == 'bf16'): the first block writes bf16, a 2-D block whose input is bf16 runs the bf16 kernels and
        hands bf16 on -- fp32 with ``out_f32`` (the last encoder block, whose consumer is the fp32 1-D stage); a block the bf16 kernels do
        not cover converts its input and continues in fp32.
        ``bn_groups`` > 1 (no-grad forward of a training-mode BatchNorm block only): the batch holds that many equal slices that the reference
        sends through the module one call after the other -- batch statistics per slice, running statistics updated slice by slice."""
        if self._is_l0_block():  # single-channel mel image: conv + norm + activation fused, output written once
            n = self.norm
            if self.norm_type == 'IN':
                return ops.L0BlockFn.apply(x_cl.squeeze(-1), self.conv.weight, None, None, None, None, None, x_cl.shape[0], self.slope,
                                           out_holder)
            return ops.L0BlockFn.apply(x_cl.squeeze(-1), self.conv.weight, n.weight, n.bias, n.running_mean, n.running_var,
                                       n.num_batches_tracked, 1, self.slope, out_holder)
        if self.norm_type == 'IN' and self.conv_type == '1d':  # conv (+ split-K reduction) + norm over C + activation
            return ops.ConvRowNormFn.apply(x_cl, self.conv.weight, self.stride, self.padding, self.slope)
        if self.conv_type == '2d' and (self.norm_type == 'IN' or self.training):
            groups = x_cl.shape[0] if self.norm_type == 'IN' else 1
            if bn_groups > 1 and self.norm_type != 'IN':
                # the fused-statistics launch accumulates ONE batch group for BatchNorm: a caller that asks for per-slice statistics must not get
                # joint ones silently (not reachable today -- the paired no-grad pass is 1-D -- hence a refusal, not a second code path)
                raise RuntimeError("bn_groups > 1 is not built for 2-D BatchNorm blocks")
            if ops.conv_stats_fusable(x_cl, self.conv.weight, self.stride, self.padding, groups):
                # the conv's epilogue accumulates the normalisation statistics: y is not re-read for them
                y, sums = ops.ConvStatsFn.apply(x_cl, self.conv.weight, self.stride, self.padding, groups, in_holder)
                n = self.norm
                if self.norm_type == 'IN':
                    return ops.ColNormActFn.apply(y, None, None, None, None, None, groups, self.slope, sums, out_holder, out_f32)
                return ops.ColNormActFn.apply(y, n.weight, n.bias, n.running_mean, n.running_var, n.num_batches_tracked, 1,
                                              self.slope, sums, out_holder, out_f32)
        if x_cl.dtype == torch.bfloat16:
            x_cl = x_cl.float()  # no bf16 kernel for this block: the rest of the chain runs in fp32
            in_holder = None
        y = ops.ConvFn.apply(x_cl, self.conv.weight, None, self.stride, self.padding, in_holder)
        if self.norm_type == 'IN':
            if self.conv_type == '2d':  # per-(b,c) statistics over H*W
                return ops.ColNormActFn.apply(y, None, None, None, None, None, y.shape[0], self.slope, None, out_holder)
            return ops.RowNormActFn.apply(y, self.slope)  # InstanceNorm1d on the permuted tensor == norm over C
        n = self.norm
        if self.training:
            if bn_groups > 1 and (torch.is_grad_enabled() and y.requires_grad):
                raise RuntimeError("bn_groups > 1 is a no-grad path (the affine gradients of a grouped launch are not built)")
            return ops.ColNormActFn.apply(y, n.weight, n.bias, n.running_mean, n.running_var, n.num_batches_tracked, int(bn_groups),
                                          self.slope, None, out_holder if self.conv_type == '2d' else None)
        if torch.is_grad_enabled() and y.requires_grad:
            raise RuntimeError("eval-mode BatchNorm is an inference-only path in this engine (wrap in torch.no_grad())")
        return ops.colnorm_eval(y, n.weight, n.bias, n.running_mean, n.running_var, self.slope)

    def forward(self, x):
        return ops.cf_view(self.forward_cl(ops.cl(x)))


def conv_head(x_cl, conv):
    """Plain nn.Conv1d with bias (generator.py:103, discriminator.py:16, autoencoder.py:56) on channels-last data."""
    return ops.ConvFn.apply(x_cl, conv.weight, conv.bias, conv.stride[0], conv.padding[0])


def make_head(cin, cout, k, stride=1, padding=0):
    conv = nn.Conv1d(cin, cout, kernel_size=k, stride=stride, padding=padding, bias=True)
    conv.weight.data = ops.to_weight_layout(conv.weight.data)
    return conv
