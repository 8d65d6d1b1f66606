"""Generator networks with the reference's class names, constructor (cfg), forward signatures and
state_dict keys (core/networks/keypoints_generation/generator.py:8-117), built from layer tables and run
channels-last through the gfx950 kernels."""
import torch
from torch import nn

from .... import ops
from ..building_blocks import ConvNormRelu, conv_head, make_head

# (cin, cout, kernel, stride, padding) of the 8-layer mel encoder, two blocks per stage (generator.py:15-30)
_SPEC_ENCODER = (
    ((1, 64, None, None, None, False), (64, 64, None, None, None, True)),
    ((64, 128, None, None, None, False), (128, 128, None, None, None, True)),
    ((128, 256, None, None, None, False), (256, 256, None, None, None, True)),
    ((256, 256, None, None, None, False), (256, 256, (6, 3), 1, 0, False)),
)
_UNET_DOWN = (False, False, True, True, True, True, True)  # e0..e6, generator.py:53-62


class AudioEncoder(nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        leaky, norm = cfg.VOICE2POSE.GENERATOR.LEAKY_RELU, cfg.VOICE2POSE.GENERATOR.NORM
        self.specgram_encoder_2d = nn.Sequential(*[
            nn.Sequential(*[ConvNormRelu('2d', ci, co, downsample=down, kernel_size=k, stride=s, padding=p, norm=norm, leaky=leaky)
                            for (ci, co, k, s, p, down) in stage])
            for stage in _SPEC_ENCODER])

    def encode_cl(self, mel):
        """mel (B,n_mels,F) -> (B,H',W',256) channels-last feature map (H'=5 for 80 mels)."""
        x = mel.unsqueeze(-1)  # (B,H,W,1): the mel image is already channels-last with C=1
        hooks = getattr(self, 'grad_bucket_hooks', None)  # {flat block index: hook on that block's INPUT gradient}
        i = 0
        holder = None  # the chain is strictly sequential: each block's output feeds exactly the next block's conv
        for stage in self.specgram_encoder_2d:
            for block in stage:
                if hooks and i in hooks and x.requires_grad:
                    # fires in backward once blocks i.. have produced their weight gradients (dp.GradReducer buckets)
                    x.register_hook(hooks[i])
                last = i == 7  # the last block feeds the resize, not a conv: nobody would use its hand-over
                nxt = ops.NormBwdHolder() if (not last and torch.is_grad_enabled()) else None
                x = block.forward_cl(x, holder, nxt, out_f32=last)
                holder = nxt if (nxt is not None and nxt.y is not None) else None
                i += 1
        return x

    def forward(self, x, num_frames):
        """(B,80,F) -> (B,256,num_frames), generator.py:39-43."""
        return ops.cf_view(ops.ResizeConcatFn.apply(self.encode_cl(x), None, int(num_frames)))


class UNet_1D(nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        leaky, norm = cfg.VOICE2POSE.GENERATOR.LEAKY_RELU, cfg.VOICE2POSE.GENERATOR.NORM
        code_dim = cfg.VOICE2POSE.GENERATOR.CLIP_CODE.DIMENSION
        for i, down in enumerate(_UNET_DOWN):
            cin = 256 + code_dim if (i == 0 and code_dim is not None) else 256
            setattr(self, 'e%d' % i, ConvNormRelu('1d', cin, 256, downsample=down, norm=norm, leaky=leaky))
        for i in (5, 4, 3, 2, 1):
            setattr(self, 'd%d' % i, ConvNormRelu('1d', 256, 256, downsample=False, norm=norm, leaky=leaky))

    def forward_cl(self, x):
        """(B,T,256[+D]) -> (B,T,256), generator.py:70-85: encoder pyramid, then conv(linear-upsample + skip)."""
        skips = []
        for i in range(7):
            x = getattr(self, 'e%d' % i).forward_cl(x)
            skips.append(x)
        for i in (5, 4, 3, 2, 1):
            skip = skips[i]
            x = getattr(self, 'd%d' % i).forward_cl(ops.UpsampleAddFn.apply(x, skip, skip.shape[1]))
        return x

    def forward(self, x):
        return ops.cf_view(self.forward_cl(ops.cl(x)))


class SequenceGeneratorCNN(nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        self.cfg = cfg
        leaky, norm = cfg.VOICE2POSE.GENERATOR.LEAKY_RELU, cfg.VOICE2POSE.GENERATOR.NORM
        self.audio_encoder = AudioEncoder(cfg)
        self.unet = UNet_1D(cfg)
        self.decoder = nn.Sequential(
            *[ConvNormRelu('1d', 256, 256, downsample=False, norm=norm, leaky=leaky) for _ in range(4)],
            make_head(256, cfg.DATASET.NUM_LANDMARKS * 2, 1))

    def _chain(self):
        """(spec, weights, slope) of the sixteen Conv1d blocks as ops.Chain1dFn takes them -- e0..e6, d5..d1 (generator.py:70-85), the four
        decoder blocks -- or None when a block is not ConvNormRelu('1d', norm='IN') (BatchNorm chains keep the per-block kernels)."""
        from .... import _lib
        blocks = [getattr(self.unet, 'e%d' % i) for i in range(7)] + [getattr(self.unet, 'd%d' % i) for i in (5, 4, 3, 2, 1)] + list(self.decoder)[:4]
        if any(b.conv_type != '1d' or b.norm_type != 'IN' or b.slope != blocks[0].slope for b in blocks):
            return None
        wiring = [(_lib.CHAIN_PLAIN, -1, -1)] + [(_lib.CHAIN_NORM, i - 1, -1) for i in range(1, 7)]
        wiring += [(_lib.CHAIN_UPADD, 6 + j, 5 - j) for j in range(5)]  # d5 = upsample(e6) + e5, d4 = upsample(d5) + e4, ...
        wiring += [(_lib.CHAIN_NORM, 11 + j, -1) for j in range(4)]
        spec = tuple((b.conv.kernel_size[0], b.stride, b.padding) + w for b, w in zip(blocks, wiring))
        return spec, [b.conv.weight for b in blocks], blocks[0].slope

    def forward(self, x, num_frames, code=None):
        """mel (B,80,F), code (B,D)|None -> poses (B,num_frames,2,K), generator.py:106-117."""
        num_frames = int(num_frames)
        feat = self.audio_encoder.encode_cl(x)
        hook = getattr(self, 'post_encoder_grad_hook', None)
        if feat.requires_grad:
            # fires in backward once the gradient w.r.t. the encoder output exists, i.e. when the U-Net / decoder backward
            # is done: their deferred weight-gradient launches go to the side stream here (ops.flush_deferred_dw), and the
            # data-parallel exchange of those gradients starts (dp.GradReducer); both overlap the much longer Conv2d
            # backward
            def _at_encoder_output(grad, _hook=hook):
                ops.stage_mark("g1d_bwd:end")
                ops.flush_deferred_dw()
                return _hook(grad) if _hook is not None else None

            feat.register_hook(_at_encoder_output)
        use_code = self.cfg.VOICE2POSE.GENERATOR.CLIP_CODE.DIMENSION is not None
        ops.stage_mark("g1d_fwd:begin")
        h = ops.ResizeConcatFn.apply(feat, code if use_code else None, num_frames)  # (B,T,256[+D])
        chain = self._chain()
        if chain is not None and ops.chain1d_usable(h, chain[0], chain[1]):
            # U-Net + decoder blocks in one persistent launch per direction (csrc/chain1d.hip)
            h = ops.Chain1dFn.apply(h, chain[0], chain[2], *chain[1])
        else:
            h = self.unet.forward_cl(h)
            for block in list(self.decoder)[:4]:
                h = block.forward_cl(h)
        h = conv_head(h, self.decoder[4])  # (B,T,2K): channel c = xy*K + k, i.e. already the (B,T,2,K) memory layout
        ops.stage_mark("g1d_fwd:end")
        if ops.STAGES is not None and h.requires_grad:
            h.register_hook(lambda g: ops.stage_mark("g1d_bwd:begin"))
        return h.reshape(-1, num_frames, 2, self.cfg.DATASET.NUM_LANDMARKS)
