#!/usr/bin/env python
"""Per-block timeline of the Conv1d chain launches (tuning build): where the 8 workgroups of a clip spend their time.
   python __graft_entry__.py --tuning && python tools/debug/chain_timeline.py [--batch 32]"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["SDT_HIP_LIB"] = os.environ.get("SDT_CHAIN_LIB") or os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speechdrivestemplates_amd import _lib, ops  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tests"))
from test_chain1d_gpu import SLOPE, _weights, _wiring  # noqa: E402

NAMES = ["e%d" % i for i in range(7)] + ["d%d" % i for i in (5, 4, 3, 2, 1)] + ["dec%d" % i for i in range(4)]


def report(tl, nwg, nsteps, names):
    t = tl[:nwg, :nsteps].astype(np.float64) * 0.01  # us
    t0 = t[:, 0, 0].min()
    print("  launch span %.1f us (first stamp -> last arrival); workgroup start skew %.2f us" % (t[:, nsteps - 1, 3].max() - t0, t[:, 0, 0].max() - t0))
    print("  %-6s %9s %9s %9s %9s %9s" % ("block", "wait", "stage", "gemm+pub", "arrive", "total"))
    prev = t[:, 0, 0]
    tot = np.zeros(4)
    for s in range(nsteps):
        w = t[:, s, 0] - prev
        st = t[:, s, 1] - t[:, s, 0]
        g = t[:, s, 2] - t[:, s, 1]
        a = t[:, s, 3] - t[:, s, 2]
        prev = t[:, s, 3]
        tot += [np.median(w), np.median(st), np.median(g), np.median(a)]
        print("  %-6s %9.2f %9.2f %9.2f %9.2f %9.2f" % (names[s], np.median(w), np.median(st), np.median(g), np.median(a), np.median(w + st + g + a)))
    print("  %-6s %9.2f %9.2f %9.2f %9.2f %9.2f   (medians over the %d workgroups)" % (("sum",) + tuple(tot) + (tot.sum(), nwg)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--math", default="f32", choices=["f32", "bf16"], help="bf16: products of bf16-rounded operands (what a bf16-storage run selects)")
    a = ap.parse_args()
    ops.CHAIN_MATH = a.math
    lib = _lib.load()
    lib.sdt_debug_set_timeline_chain.argtypes = [ctypes.c_void_p]
    B = a.batch
    ws = _weights(288, 1)
    h = torch.randn((B, 64, 288), device="cuda", requires_grad=True)
    gz = torch.randn((B, 64, 256), device="cuda")
    nwg = 64 * ((B + 7) // 8)
    for it in range(3):
        ops.begin_step(torch.device("cuda", 0))
        buf_f = torch.zeros((nwg, 24, 4), dtype=torch.int64, device="cuda")
        buf_b = torch.zeros((nwg, 24, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        assert lib.sdt_debug_set_timeline_chain(ctypes.c_void_p(buf_f.data_ptr())) == 0
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        z = ops.Chain1dFn.apply(h, _wiring(), SLOPE, *ws)
        e[1].record()
        torch.cuda.synchronize()
        assert lib.sdt_debug_set_timeline_chain(ctypes.c_void_p(buf_b.data_ptr())) == 0
        ops.defer_small_dw(True)
        e[2].record()
        z.backward(gz)
        e[3].record()
        ops.defer_small_dw(False)
        ops.join_side_stream()
        torch.cuda.synchronize()
        lib.sdt_debug_set_timeline_chain(ctypes.c_void_p(0))
    print("B = %d: forward launch %.1f us, backward (chain launch + enqueue of the deferred weight gradients) %.1f us (HIP events)" % (B, e[0].elapsed_time(e[1]) * 1e3, e[2].elapsed_time(e[3]) * 1e3))
    used = [w for w in range(nwg) if ((w >> 6) * 8 + (w & 7)) < B]
    print("forward:")
    report(buf_f.cpu().numpy()[used], len(used), 16, NAMES)
    print("backward:")
    report(buf_b.cpu().numpy()[used], len(used), 16, NAMES[::-1])


if __name__ == "__main__":
    main()
