#!/usr/bin/env python
"""conv1d_small_kernel (tuning library: sdt_debug_set_small1d) against conv_taps_kernel on the 1-D stage's shapes: bit identity and time."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["SDT_HIP_LIB"] = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
import torch  # noqa: E402

from speechdrivestemplates_amd import _lib, ops  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
B = 32
for (T, Cin, Cout, k, s, p) in [(64, 256, 256, 3, 1, 1), (64, 288, 256, 3, 1, 1), (64, 256, 256, 4, 2, 1), (32, 256, 256, 3, 1, 1), (16, 256, 256, 4, 2, 1),
                                (8, 256, 256, 3, 1, 1), (2, 256, 256, 3, 1, 1), (37, 256, 256, 3, 1, 1), (64, 64, 128, 3, 1, 1), (64, 242, 64, 3, 1, 1)]:
    x = torch.randn((B, T, Cin), device="cuda")
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn((Cout, Cin, k), device="cuda") * 0.05))
    res = {}
    for on in (0, 1):
        lib.sdt_debug_set_small1d(ctypes.c_int(on))
        y = ops.conv_forward(x, w, None, s, p)
        gy = torch.randn_like(y) if on == 0 else gy
        dx = ops.conv_input_grad(gy, w, x.shape, s, p)
        torch.cuda.synchronize()
        ts = []
        for fn in (lambda: ops.conv_forward(x, w, None, s, p), lambda: ops.conv_input_grad(gy, w, x.shape, s, p)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 20)
        res[on] = (y, dx, ts)
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    print("T %3d %3d->%3d k%d s%d: bit-identical %s   fwd %.1f -> %.1f us   dX %.1f -> %.1f us (incl. split-K reduce)" % (
        T, Cin, Cout, k, s, same, res[0][2][0], res[1][2][0], res[0][2][1], res[1][2][1]))
