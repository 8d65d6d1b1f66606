#!/usr/bin/env python
"""Are the fp64-atomic normalisation statistics order-independent in practice?  The same conv + statistics epilogue (and the same input
gradient + backward statistics) is launched many times; every launch's fp64 sums are compared BITWISE with the first launch's.  (DESIGN.md
section 2, Normalisation: fp64 sums of fp32 partials of similar magnitude are exact, hence independent of the arrival order.)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402
from conv_bench import LAYERS  # noqa: E402

torch.manual_seed(0)
B, reps = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 40
tot_stats = tot_diff = 0
for name, Hi, Wi, Cin, Cout, kh, kw, s, p in LAYERS:
    if Hi == 1 or Cin == 1:
        continue
    x = torch.randn((B, Hi, Wi, Cin), device="cuda")
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn((Cout, Cin, kh, kw), device="cuda") * 0.05))
    first_f = first_b = None
    nd_f = nd_b = 0
    for r in range(reps):
        ops.begin_step()
        y, sums = ops.ConvStatsFn.apply(x, w, s, p, B)
        if r == 0:
            gy = torch.randn_like(y)
            h0 = dict(y=torch.randn_like(x), mean=torch.randn((B, Cin), device="cuda") * 0.1, rstd=torch.rand((B, Cin), device="cuda") + 0.5)
        h = ops.NormBwdHolder()
        h.y, h.mean, h.rstd, h.gamma, h.beta, h.groups, h.slope = h0["y"], h0["mean"], h0["rstd"], None, None, B, 0.2
        ops.conv_input_grad(gy, w, x.shape, s, p, h)
        torch.cuda.synchronize()
        f = sums.clone().view(torch.int64)
        b = h.sums.clone().view(torch.int64) if h.sums is not None else None
        if first_f is None:
            first_f, first_b = f, b
        else:
            nd_f += int((f != first_f).sum())
            nd_b += int((b != first_b).sum()) if b is not None else 0
    n = first_f.numel() + (first_b.numel() if first_b is not None else 0)
    tot_stats += n * (reps - 1)
    tot_diff += nd_f + nd_b
    print("%s: %d statistics x %d repeats: forward sums differing bitwise from the first launch %d, backward %d" % (name, n, reps - 1, nd_f, nd_b))
print("total: %d of %d (statistic, launch) pairs differ in their fp64 bits" % (tot_diff, tot_stats))
