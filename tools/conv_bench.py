#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM conv kernels on the hot path's layer shapes (B=32 by default).
   python tools/conv_bench.py [--only L2] [--reps 20] [--roles fwd,dX,dW]
The SDT_CONV_PRIO / SDT_CONV_TILE / SDT_DW_TILE experiment switches exist only in the -DSDT_TUNING library: build it with
`python __graft_entry__.py --tuning` and run this tool with --tuning (or SDT_HIP_LIB=.../libsdt_hip_tuning.so)."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if "--tuning" in sys.argv:  # must be set before the package binds the library
    os.environ["SDT_HIP_LIB"] = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
import torch  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402

LAYERS = [  # name, Hi, Wi, Cin, Cout, kh, kw, s, p
    ("L0", 80, 427, 1, 64, 3, 3, 1, 1), ("L1", 80, 427, 64, 64, 4, 4, 2, 1), ("L2", 40, 213, 64, 128, 3, 3, 1, 1),
    ("L3", 40, 213, 128, 128, 4, 4, 2, 1), ("L4", 20, 106, 128, 256, 3, 3, 1, 1), ("L5", 20, 106, 256, 256, 4, 4, 2, 1),
    ("L6", 10, 53, 256, 256, 3, 3, 1, 1), ("L7", 10, 53, 256, 256, 6, 3, 1, 0),
    ("c1d_k3_T64", 1, 64, 256, 256, 1, 3, 1, 1), ("c1d_k4s2_T64", 1, 64, 256, 256, 1, 4, 2, 1), ("c1d_k3_T8", 1, 8, 256, 256, 1, 3, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--tuning", action="store_true", help="load the -DSDT_TUNING library (experiment switches live there)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--roles", default="fwd,dX,dW")
    ap.add_argument("--math", default="f32", help="f32 | bf16 | bf16x3 | bf16x6 (forward / input-gradient kernels)")
    ap.add_argument("--oob", action="store_true", help="experiment: the stream-K kernel's A / B loads all out of range (zeros, no memory traffic): the loop's issue-bound rate")
    ap.add_argument("--streamk", type=int, default=2, help="0: the 64x64 kernel of conv.hip; 1 / 2: the persistent stream-K kernel with 1 / 2 workgroups per CU")
    a = ap.parse_args()
    ops.USE_STREAMK = a.streamk > 0
    ops.USE_STREAMK_DW = a.streamk > 0
    if os.environ.get("SDT_SK_ALL") == "1":  # every 2-D layer through the stream-K kernel (tool-only switch)
        ops.STREAMK_MIN_STEPS, ops.STREAMK_MIN_COUT = 1, 64
    if a.oob:
        def _oob_launch(plan, x4, ws_w, bias, y, stats, nb, st):
            lib = ops._lib.load()
            wsb, epoch = ops._sk_workspace(y.device, st), 1
            return lib.sdt_convsk_f32(ops._p(x4), ops._p(ws_w), ops._p(bias), ops._p(y), plan.host, ops._p(plan.dev), ops._p(wsb), epoch, ops._p(stats), nb,
                                      16, 16, y.numel() * 4, st)
        ops._sk_launch = _oob_launch
    if a.streamk > 0:
        from speechdrivestemplates_amd import _lib
        _lib.check(_lib.load().sdt_convsk_set_wg_per_cu(a.streamk))
    if os.environ.get("SDT_SK_NTMAJOR"):  # tuning library: weight bytes above which a stream-K plan orders its tiles n-tile major
        import ctypes
        _lib.check(_lib.load().sdt_convsk_set_ntmajor_bytes(ctypes.c_int(int(os.environ["SDT_SK_NTMAJOR"]))))
    B = a.batch
    roles = a.roles.split(",")
    for name, Hi, Wi, Cin, Cout, kh, kw, s, p in LAYERS:
        if a.only and name not in a.only.split(","):
            continue
        one_d = Hi == 1
        x = torch.randn((B, Wi, Cin) if one_d else (B, Hi, Wi, Cin), device="cuda")
        wshape = (Cout, Cin, kw) if one_d else (Cout, Cin, kh, kw)
        w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(wshape, device="cuda") * 0.05))
        y = ops.conv_forward(x, w, None, s, p)
        gy = torch.randn_like(y)
        flops = 2.0 * y.numel() * Cin * kh * kw
        def holder(groups):
            h = ops.NormBwdHolder()
            h.y = torch.randn_like(x)
            h.mean = torch.zeros((groups, Cin), device="cuda")
            h.rstd = torch.ones((groups, Cin), device="cuda")
            h.gamma, h.beta = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
            h.groups, h.slope = groups, 0.2
            return h
        hs = {1: holder(1), B: holder(B)}
        fns = {"fwd": lambda: ops.conv_forward(x, w, None, s, p),
               # + the statistics epilogues of the step: BatchNorm (one group) / InstanceNorm (one group per item)
               "fwdS1": lambda: ops.ConvStatsFn.apply(x, w, s, p, 1), "fwdSB": lambda: ops.ConvStatsFn.apply(x, w, s, p, B),
               "dXS1": lambda: ops.conv_input_grad(gy, w, x.shape, s, p, hs[1]), "dXSB": lambda: ops.conv_input_grad(gy, w, x.shape, s, p, hs[B]),
               "dX": lambda: ops.conv_input_grad(gy, w, x.shape, s, p),
               "dW": lambda: ops.conv_weight_grad(x, gy, w, s, p)}
        for role in roles:
            if (role.startswith("dX") and name == "L0") or (role not in ("fwd", "dX", "dW") and one_d):
                continue
            fn = fns[role]
            if role not in ("fwd", "dX", "dW"):
                fn = (lambda f: lambda: (ops.begin_step(), f())[1])(fns[role])  # recycles the zeroed statistics arena (one small fill)
            err = ""
            if a.math != "f32":
                def run():
                    if role != "dW":
                        return fn()
                    w.grad = None
                    fn()
                    return w.grad.clone()
                ref = run()
                ops.set_conv_math(a.math)
                got = run()
                err = "  max|d|/max|ref| vs f32 kernel %.2e" % ((got - ref).abs().max() / ref.abs().max()).item()
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            ops.set_conv_math("f32")
            print("%-14s %-5s  %8.1f us  %6.1f TFLOP/s  (%.1f GFLOP, out %s)%s" % (name, role, us, flops / us / 1e6, flops / 1e9, tuple(y.shape), err), flush=True)


if __name__ == "__main__":
    main()
