#!/usr/bin/env python
"""Weight gradient of the ordered stream-K kernel (convsk_dw_kernel, every tile shape) against the slab kernel of conv.hip and float64."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402

torch.manual_seed(0)
for (B, H, W, Cin, Cout, k, s, p) in [(32, 40, 213, 64, 64, 3, 1, 1), (4, 40, 213, 64, 64, 3, 1, 1), (4, 80, 427, 64, 64, 4, 2, 1), (4, 40, 213, 64, 128, 3, 1, 1),
                                      (4, 40, 213, 128, 128, 4, 2, 1), (8, 20, 106, 128, 256, 3, 1, 1), (4, 20, 106, 64, 192, 3, 1, 1), (3, 33, 77, 192, 64, 3, 1, 1)]:
    x = torch.randn((B, H, W, Cin), device="cuda")
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn((Cout, Cin, k, k), device="cuda") * 0.05))
    y = ops.conv_forward(x, w, None, s, p)
    gy = torch.randn_like(y)
    out = {}
    for sk in (False, True):
        ops.USE_STREAMK_DW = sk
        w.grad = None
        ops.conv_weight_grad(x, gy, w, s, p)
        out[sk] = w.grad.clone()
    xd = x.permute(0, 3, 1, 2).double()
    wd = w.detach().double().contiguous().requires_grad_(True)  # logical shape (Cout, Cin, kh, kw)
    F.conv2d(xd, wd, None, s, p).backward(gy.permute(0, 3, 1, 2).double())
    ref = wd.grad
    g = ops.conv_geom_for(x.shape, w, s, p)
    from speechdrivestemplates_amd import _lib
    sup = _lib.load().sdt_convsk_dw_supported(g)
    e = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()
    print("B%d %dx%d %d->%d k%d s%d: streamk supported %d   sk vs slab %.2e%s" % (B, H, W, Cin, Cout, k, s, sup, e(out[True], out[False]),
          "" if ref is None else "   sk vs f64 %.2e   slab vs f64 %.2e" % (e(out[True], ref), e(out[False], ref))))
