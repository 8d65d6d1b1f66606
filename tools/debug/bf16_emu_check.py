#!/usr/bin/env python
"""Where does the bf16-storage path round?  One 2-D block (conv with statistics + normalise + LeakyReLU) and the first block on bf16 storage against the
float64 emulation of oracle.sdt_oracle.BF16_EMULATION: fraction of stored bf16 values that differ, and by how many bf16 ulps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402


def rb(t):
    return t.to(torch.bfloat16).to(t.dtype)


def cmp(name, got_bf, ref64):
    """got: bf16 tensor (device), ref64: float64 values BEFORE the final rounding"""
    g = got_bf.float().double().cpu()
    r = rb(ref64)
    diff = (g != r)
    ulp = (ref64.abs().clamp_min(1e-30).log2().floor() - 7).exp2()
    print("  %-34s differing %.2e of %d   max |diff| %.2f ulp   rel-max-err %.2e (unrounded ref %.2e)"
          % (name, diff.double().mean().item(), g.numel(), ((g - r).abs() / ulp).max().item(), ((g - r).abs().max() / r.abs().max()).item(),
             ((g - ref64).abs().max() / ref64.abs().max()).item()))


def main():
    dev = torch.device("cuda", 0)
    ops.set_storage("bf16")
    B = 8
    g = torch.Generator().manual_seed(1)
    # ---- block 0
    mel = torch.randn(B, 80, 427, generator=g) * 2.0 - 4.0
    w0 = torch.randn(64, 1, 3, 3, generator=g) * 0.3
    w0p = torch.nn.Parameter(ops.to_weight_layout(w0).to(dev))
    z0 = ops.L0BlockFn.apply(mel.to(dev), w0p, None, None, None, None, None, B, 0.2, None)
    y = F.conv2d(mel.double().unsqueeze(1), w0.double(), None, 1, 1)
    ref = F.leaky_relu(F.instance_norm(y, eps=1e-5), 0.2).permute(0, 2, 3, 1)
    cmp("block 0 output", z0, ref)
    # ---- a 2-D block on the bf16 output of block 0
    for tag, Cin, Cout, k, s, p in (("L1", 64, 64, 4, 2, 1), ("L2", 64, 128, 3, 1, 1)):
        x = z0 if tag == "L1" else z1
        w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
        wp = torch.nn.Parameter(ops.to_weight_layout(w).to(dev))
        ops._ARENA.begin_step(dev)
        yb, sums = ops.ConvStatsFn.apply(x, wp, s, p, B, None)
        zb = ops.ColNormActFn.apply(yb, None, None, None, None, None, B, 0.2, sums, None, False)
        x64 = x.float().double().cpu().permute(0, 3, 1, 2)
        y64 = F.conv2d(x64, rb(w.double()), None, s, p)
        cmp(tag + " conv output as stored", yb, y64.permute(0, 2, 3, 1))
        mean, var = y64.mean((2, 3), keepdim=True), y64.var((2, 3), unbiased=False, keepdim=True)
        z64 = F.leaky_relu((rb(y64) - mean) / torch.sqrt(var + 1e-5), 0.2)
        cmp(tag + " activated output", zb, z64.permute(0, 2, 3, 1))
        # the same from the kernel's OWN stored conv output (isolates the normalisation pass)
        yk = yb.float().double().cpu().permute(0, 3, 1, 2)
        z64k = F.leaky_relu((yk - mean) / torch.sqrt(var + 1e-5), 0.2)
        cmp(tag + " activated (kernel's own y)", zb, z64k.permute(0, 2, 3, 1))
        z1 = zb
    ops.set_storage("f32")


if __name__ == "__main__":
    main()
