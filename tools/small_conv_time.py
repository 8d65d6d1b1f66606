#!/usr/bin/env python
"""GPU time per launch of the small (1-D stage) conv launches, measured as back-to-back launches on one stream between two
events -- i.e. kernel duration + the dispatch gap a dependent chain pays.  python tools/small_conv_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speechdrivestemplates_amd import _lib, ops  # noqa: E402
from speechdrivestemplates_amd.ops import _p  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = 32
for T, k, s in ((64, 3, 1), (64, 4, 2), (16, 3, 1), (4, 3, 1)):
    x = torch.randn(B, T, 256, device="cuda")
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(256, 256, k, device="cuda") * 0.05))
    x4 = x.unsqueeze(1)
    g = ops.conv_geom_for(x4.shape, w, s, 1)
    kk = lib.sdt_conv_taps_splitk_hint(g)
    y = torch.empty((g.B, g.Ho, g.Wo, g.Cout), device="cuda")
    part = torch.empty((kk,) + tuple(y.shape), device="cuda")
    ws = ops.weight_storage(w)
    z, mean, rstd = torch.empty_like(y), torch.empty(g.B * g.Wo, device="cuda"), torch.empty(g.B * g.Wo, device="cuda")

    def conv():
        lib.sdt_conv_taps_splitk_f32(_p(x4), _p(ws), None, _p(y), g, kk, _p(part), st)

    def norm():
        lib.sdt_rownorm_slabs_fwd_f32(_p(part), kk, _p(y), _p(z), _p(mean), _p(rstd), g.B * g.Wo, 256, 1e-5, 0.2, st)

    for name, fn in (("conv", conv), ("rownorm+reduce", norm)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 300
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("T=%3d k%d s%d  splitk %2d  grid %4d WGs  %-15s %6.2f us/launch" % (T, k, s, kk, -(-g.B * g.Wo // 64) * 4 * kk, name, e0.elapsed_time(e1) * 1e3 / n))
