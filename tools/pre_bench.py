#!/usr/bin/env python
"""Micro-benchmark of sdt_conv_taps_pre_f32 (bf16x6 products from pre-split planes) on the audio-encoder layer shapes at B=32:
forward and input gradient, tile 64x64 / 128x64 / 128x128, next to the exact-fp32 MFMA kernel.
   python tools/pre_bench.py [--only L2,L5] [--reps 10]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from speechdrivestemplates_amd import _lib, ops  # noqa: E402

LAYERS = [("L1", 80, 427, 64, 64, 4, 4, 2, 1), ("L2", 40, 213, 64, 128, 3, 3, 1, 1), ("L3", 40, 213, 128, 128, 4, 4, 2, 1),
          ("L4", 20, 106, 128, 256, 3, 3, 1, 1), ("L5", 20, 106, 256, 256, 4, 4, 2, 1), ("L6", 10, 53, 256, 256, 3, 3, 1, 1),
          ("L7", 10, 53, 256, 256, 6, 3, 1, 0)]


def planes(t):
    t = t.contiguous()
    p = ops.planes_like(t)
    _lib.check(_lib.load().sdt_split_planes_f32(t.data_ptr(), p.data_ptr(), t.numel(), t.shape[-1], torch.cuda.current_stream().cuda_stream))
    return p


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    lib, B = _lib.load(), a.batch
    st = torch.cuda.current_stream().cuda_stream
    for name, Hi, Wi, Cin, Cout, kh, kw, s, p in LAYERS:
        if a.only and name not in a.only.split(","):
            continue
        x = torch.randn(B, Hi, Wi, Cin, device="cuda")
        w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(Cout, Cin, kh, kw, device="cuda") * 0.05))
        ws = ops.weight_storage(w.data)
        wt = ws.permute(2, 1, 0).contiguous()
        geo = ops.fwd_geom(B, Hi, Wi, Cin, Cout, kh, kw, s, p)
        y = torch.empty((B, geo.Ho, geo.Wo, Cout), device="cuda")
        gy = torch.randn_like(y)
        dx = torch.empty_like(x)
        xp, wp, wtp, gyp = planes(x), planes(ws), planes(wt), planes(gy)
        arr, n, gs = ops.dx_pack(B, Hi, Wi, Cin, Cout, kh, kw, s, p, False)
        flops = 2.0 * y.numel() * Cin * kh * kw
        yref = ops.conv_forward(x, w, None, s, p)
        t_f = timeit(lambda: ops.conv_forward(x, w, None, s, p), a.reps)
        t_d = timeit(lambda: ops.conv_input_grad(gy, w, x.shape, s, p), a.reps)
        line = "%-3s fp32-MFMA fwd %7.1f us %6.1f TF | dX %7.1f us %6.1f TF ||" % (name, t_f, flops / t_f / 1e6, t_d, flops / t_d / 1e6)
        for tile in (64064, 128128, 1281288, 1282568):
            _lib.check(lib.sdt_set_pre_tile(tile))
            f = lambda: _lib.check(lib.sdt_conv_taps_pre_f32(xp.data_ptr(), xp.shape[1], wp.data_ptr(), wp.shape[1], y.data_ptr(), geo, 1, None, 0, None, st))  # noqa: E731
            d = lambda: _lib.check(lib.sdt_conv_taps_pre_f32(gyp.data_ptr(), gyp.shape[1], wtp.data_ptr(), wtp.shape[1], dx.data_ptr(), arr, n, None, 0, None, st))  # noqa: E731
            tf, td = timeit(f, a.reps), timeit(d, a.reps)
            err = ((y - yref).abs().max() / yref.abs().max()).item()
            line += " pre%d fwd %6.1f us %6.1f TF dX %6.1f us %6.1f TF (err %.0e) |" % (tile, tf, flops / tf / 1e6, td, flops / td / 1e6, err)
        _lib.check(lib.sdt_set_pre_tile(0))
        print(line, flush=True)


if __name__ == "__main__":
    main()
