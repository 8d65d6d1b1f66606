#!/usr/bin/env python
"""The split-fp32 form of the 8-wave conv kernel (csrc/convbf.hip, ET = float) against the fp32-MFMA kernels and a float64 reference:
forward (+ statistics) and input gradient of every Conv2d layer shape of the audio encoder.   python tools/debug/x3_check.py [--batch 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402

LAYERS = [("L1", 80, 427, 64, 64, 4, 4, 2, 1), ("L2", 40, 213, 64, 128, 3, 3, 1, 1), ("L3", 40, 213, 128, 128, 4, 4, 2, 1),
          ("L4", 20, 106, 128, 256, 3, 3, 1, 1), ("L5", 20, 106, 256, 256, 4, 4, 2, 1), ("L6", 10, 53, 256, 256, 3, 3, 1, 1),
          ("L7", 10, 53, 256, 256, 6, 3, 1, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--layers", default=",".join(c[0] for c in LAYERS))
    a = ap.parse_args()
    B, dev = a.batch, torch.device("cuda", 0)
    worst = 0.0
    for tag, Hi, Wi, Cin, Cout, kh, kw, s, p in LAYERS:
        if tag not in a.layers.split(","):
            continue
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, Hi, Wi, Cin, generator=g)
        wl = torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wl.double(), None, s, p).permute(0, 2, 3, 1).contiguous()
        gy = torch.randn(ref.shape, generator=g)
        ref_dx = torch.nn.grad.conv2d_input((B, Cin, Hi, Wi), wl.double(), gy.permute(0, 3, 1, 2).double(), s, p).permute(0, 2, 3, 1).contiguous()
        xd, gyd = x.to(dev), gy.to(dev)
        w = torch.nn.Parameter(ops.to_weight_layout(wl).to(dev))
        out = {}
        for split in (0, 1):
            ops.F32_SPLIT = bool(split)
            ops.clear_plans()
            ops._ARENA.begin_step(dev)
            y, sums = ops.ConvStatsFn.apply(xd, w, s, p, B, None)
            y0 = ops.conv_forward(xd, w, None, s, p)
            dx = ops.conv_input_grad(gyd, w, xd.shape, s, p, None)
            torch.cuda.synchronize()
            out[split] = (y.double().cpu(), sums.double().cpu().clone(), dx.double().cpu(), y0.double().cpu())
        rms_y, rms_dx = ref.pow(2).mean().sqrt().item(), ref_dx.pow(2).mean().sqrt().item()
        e = {k: ((out[k][0] - ref).abs().max().item() / rms_y, (out[k][2] - ref_dx).abs().max().item() / rms_dx, (out[k][3] - ref).abs().max().item() / rms_y,
                 (out[k][0] - ref).pow(2).mean().sqrt().item() / rms_y, (out[k][2] - ref_dx).pow(2).mean().sqrt().item() / rms_dx) for k in (0, 1)}
        # statistics: per-(clip, channel) sum and sum of squares of the stored output
        rs = torch.stack([ref.reshape(B, -1, Cout).sum(1), ref.reshape(B, -1, Cout).pow(2).sum(1)], -1).reshape(-1)
        es = {k: ((out[k][1].reshape(-1) - rs).abs().max().item() / rs.abs().max().item()) for k in (0, 1)}
        print("%s  max|err|/rms  fwd+stats: fp32-MFMA %.2e  split %.2e | fwd: %.2e  %.2e | dX: %.2e  %.2e || rms err fwd %.2e  %.2e  dX %.2e  %.2e || sums %.1e  %.1e"
              % (tag, e[0][0], e[1][0], e[0][2], e[1][2], e[0][1], e[1][1], e[0][3], e[1][3], e[0][4], e[1][4], es[0], es[1]))
        worst = max(worst, e[1][0] / max(e[0][0], 1e-12), e[1][1] / max(e[0][1], 1e-12))
    print("worst split / fp32-MFMA error ratio %.2f" % worst)
    assert not ops.streamk_error_codes()


if __name__ == "__main__":
    main()
