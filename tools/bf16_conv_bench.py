#!/usr/bin/env python
"""Per-layer timing of the bf16-storage conv kernels (forward with statistics, input gradient with backward statistics, weight gradient) at
B = 32, each launch alone on the GPU: HIP events over REP back-to-back launches.  python tools/bf16_conv_bench.py [--layers L1,L5] [--rep 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402

LAYERS = [("L1", 80, 427, 64, 64, 4, 4, 2, 1), ("L2", 40, 213, 64, 128, 3, 3, 1, 1), ("L3", 40, 213, 128, 128, 4, 4, 2, 1),
          ("L4", 20, 106, 128, 256, 3, 3, 1, 1), ("L5", 20, 106, 256, 256, 4, 4, 2, 1), ("L6", 10, 53, 256, 256, 3, 3, 1, 1),
          ("L7", 10, 53, 256, 256, 6, 3, 1, 0)]


def timed(fn, rep):
    """mean GPU time of one call: REP calls captured into a hipGraph and replayed (enqueued from Python a call costs the host ~100 us, more than
    the kernels take), HIP events around the replays"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    ops.prepare_capture_stream(torch.device("cuda", 0), st)
    with torch.cuda.stream(st):
        g.capture_begin()
        for _ in range(rep):
            fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(st)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * rep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default=",".join(c[0] for c in LAYERS))
    ap.add_argument("--rep", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-dw", action="store_true", help="skip the weight-gradient launches")
    ap.add_argument("--split", type=int, default=1, help="--dtype f32: 1 = the split-fp32 form of the 8-wave kernel (ops.F32_SPLIT), 0 = the fp32-MFMA kernels")
    ap.add_argument("--old", action="store_true", help="the round-4 128-row bf16 kernels (ops.BF16_SHAPED = False) instead of the bf16-shaped ones")
    ap.add_argument("--w3", type=int, default=1, help="--dtype f32 split: 1 = weights pre-split into three bf16 planes once (ops.W3_PRESPLIT; the weights are registered "
                                                     "with a WeightMirrors group as an optimiser would), 0 = split by every tile's loader")
    ap.add_argument("--dw-wide", type=int, default=0, help="--dtype f32 split: tile rule of the weight gradient (ops.SK_DW_WIDE)")
    ap.add_argument("--korder", type=int, default=None, help="K order of the 8-wave kernels' tiles: 0 tap-major, 1 chunk-major (ops.SK_K_ORDER)")
    a = ap.parse_args()
    if a.korder is not None:
        ops.SK_K_ORDER = a.korder
    B, dev = a.batch, "cuda"
    ops.W3_PRESPLIT = bool(a.w3)
    ops.SK_DW_WIDE = bool(a.dw_wide)
    keep = []
    ops.BF16_SHAPED = not a.old
    ops.F32_SPLIT = bool(a.split)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    tot = {"fwd": 0.0, "dX": 0.0, "dW": 0.0}
    print("%-4s %10s | %8s %8s | %8s %8s | %8s %8s   (us, TFLOP/s; %s tensors, B=%d)" % ("", "GFLOP", "fwd", "", "dX", "", "dW", "", a.dtype, B))
    for tag, Hi, Wi, Cin, Cout, kh, kw, s, p in LAYERS:
        if tag not in a.layers.split(","):
            continue
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, Hi, Wi, Cin, generator=g).to(dt).to(dev)
        w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev))
        if a.dtype == "f32" and a.w3:
            keep.append(ops.WeightMirrors([w]))  # transposed mirror + pre-split planes, refreshed on first use (as optim.FlatAdam keeps them)
        y, sums = ops.ConvStatsFn.apply(x, w, s, p, B, None)
        gy = torch.randn(y.shape, generator=g).to(dt).to(dev)
        geo = ops.conv_geom_for(x.shape, w, s, p)
        gflop = 2.0 * B * geo.Ho * geo.Wo * Cout * kh * kw * Cin / 1e9
        # the block below: a holder with the statistics of a tensor of x's shape (what the input-gradient epilogue reads back)
        h = ops.NormBwdHolder()
        h.y, h.mean, h.rstd = x, torch.zeros(B * Cin, device=dev), torch.ones(B * Cin, device=dev)
        h.groups, h.slope = B, 0.2

        def f_fwd():
            ops._ARENA.begin_step(torch.device(dev, 0))
            ops.ConvStatsFn.apply(x, w, s, p, B, None)

        def f_dx():
            ops._ARENA.begin_step(torch.device(dev, 0))
            ops.conv_input_grad(gy, w, x.shape, s, p, h if tag != "L1" else None)

        def f_dw():
            ops.conv_weight_grad(x, gy, w, s, p)
        t = {"fwd": timed(f_fwd, a.rep), "dX": timed(f_dx, a.rep), "dW": 1e-9 if a.no_dw else timed(f_dw, a.rep)}
        for k in tot:
            tot[k] += t[k]
        print("%-4s %10.2f | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f" % (tag, gflop, t["fwd"], gflop / t["fwd"] * 1e3, t["dX"], gflop / t["dX"] * 1e3,
                                                                  t["dW"], gflop / t["dW"] * 1e3))
    print("sum  fwd %.0f us  dX %.0f us  dW %.0f us  = %.3f ms" % (tot["fwd"], tot["dX"], tot["dW"], sum(tot.values()) / 1e3))
    assert os.environ.get("SDT_ALLOW_NAN") or not ops.streamk_error_codes()


if __name__ == "__main__":
    main()
