#!/usr/bin/env python
"""Normalisation-backward statistics of the input-gradient epilogues (convsk EPI 2 and conv_taps EPI 2) against float64, groups = 1 / B."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402

torch.manual_seed(0)
slope = 0.2
for (B, H, W, Cin, Cout, kh, kw, s, p) in [(4, 20, 106, 128, 256, 3, 3, 1, 1), (4, 10, 53, 256, 256, 3, 3, 1, 1), (4, 10, 53, 256, 256, 6, 3, 1, 0),
                                          (32, 10, 53, 256, 256, 6, 3, 1, 0), (4, 20, 106, 128, 256, 4, 4, 2, 1), (4, 40, 213, 64, 128, 3, 3, 1, 1)]:
    x = torch.randn((B, H, W, Cin), device="cuda")
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn((Cout, Cin, kh, kw), device="cuda") * 0.05))
    y = ops.conv_forward(x, w, None, s, p)
    gy = torch.randn_like(y)
    for groups in (1, B):
        for sk in (False, True):
            ops.USE_STREAMK = sk
            ops._SK_PLANS.clear()
            h = ops.NormBwdHolder()
            h.y = torch.randn_like(x)
            h.mean = torch.randn((groups, Cin), device="cuda") * 0.1
            h.rstd = torch.rand((groups, Cin), device="cuda") + 0.5
            h.gamma, h.beta = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
            h.groups, h.slope = groups, slope
            ops.begin_step()
            dx = ops.conv_input_grad(gy, w, x.shape, s, p, h)
            torch.cuda.synchronize()
            if h.sums is None:
                print("B%d %dx%d %d->%d k%dx%d s%d groups %d streamk %d: statistics not fused" % (B, H, W, Cin, Cout, kh, kw, s, groups, sk))
                continue
            sums = h.sums.view(groups, Cin, 2).clone()
            dxd = dx.double().view(groups, -1, Cin)
            yh = (h.y.double().view(groups, -1, Cin) - h.mean.double()[:, None]) * h.rstd.double()[:, None]
            pre = yh * h.gamma.double() + h.beta.double()
            gg = dxd * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope))
            ref = torch.stack([gg.sum(1), (gg * yh).sum(1)], -1)
            err = ((sums - ref).abs().amax((0, 1)) / ref.abs().amax((0, 1))).tolist()
            print("B%d %dx%d %d->%d k%dx%d s%d groups %d streamk %d: sums rel err %.2e %.2e" % (B, H, W, Cin, Cout, kh, kw, s, groups, sk, err[0], err[1]))
