#!/usr/bin/env python
"""Per-workgroup timeline of conv_taps_kernel launches (tuning build, SDT_CONV_PRIO=30): every workgroup stamps the 100 MHz
real-time counter at entry / prologue done / first tile in LDS / K loop done / epilogue issued, plus its hardware id.

    python __graft_entry__.py --tuning
    SDT_CONV_PRIO=30 python tools/debug/taps_timeline.py [--only L2,L5] [--roles fwd,dX]

Prints per launch: span, workgroups per CU (min / max / histogram), the phases of a workgroup's life (median, p90), how many
workgroups are in their K loop over time (pipe cover), and the idle tail (last 10 % of the span)."""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["SDT_HIP_LIB"] = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
os.environ.setdefault("SDT_CONV_PRIO", "30")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speechdrivestemplates_amd import _lib, ops  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tools"))
from conv_bench import LAYERS  # noqa: E402


def analyse(tl, name, role, flops):
    tl = tl[tl[:, 0] != 0]
    n = len(tl)
    t = tl[:, :5].astype(np.float64) * 0.01  # us
    t0 = t[:, 0].min()
    t -= t0
    span = t[:, 4].max()
    hw = tl[:, 5]
    xcc = (hw >> 32) & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ids, cnt = np.unique(cuid, return_counts=True)
    hist = dict(zip(*np.unique(cnt, return_counts=True)))
    print("%s %s: %d workgroups, span %.1f us, %.1f TFLOP/s over the span; %d distinct CUs, workgroups/CU min %d max %d mean %.2f  histogram %s"
          % (name, role, n, span, flops / span / 1e6, len(ids), cnt.min(), cnt.max(), cnt.mean(), {int(k): int(v) for k, v in hist.items()}))
    ph = {"prologue": t[:, 1] - t[:, 0], "first tile (load -> LDS)": t[:, 2] - t[:, 1], "K loop": t[:, 3] - t[:, 2],
          "epilogue issue": t[:, 4] - t[:, 3], "whole life": t[:, 4] - t[:, 0]}
    for k, v in ph.items():
        print("    %-26s median %7.2f us   p10 %7.2f   p90 %7.2f   max %7.2f" % (k, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max()))
    # start-time clustering: how many "generations"
    starts = np.sort(t[:, 0])
    edges = np.linspace(0, span, 41)
    in_loop = [(np.sum((t[:, 2] <= x) & (t[:, 3] > x))) for x in (edges[:-1] + edges[1:]) / 2]
    alive = [(np.sum((t[:, 0] <= x) & (t[:, 4] > x))) for x in (edges[:-1] + edges[1:]) / 2]
    print("    workgroups in their K loop / alive, 40 slices of the span (per CU):")
    print("      " + " ".join("%.1f" % (v / 256.0) for v in in_loop))
    print("      " + " ".join("%.1f" % (v / 256.0) for v in alive))
    # per-CU busy: time with >=1 workgroup in K loop on that CU
    last_end = np.array([t[cuid == c, 4].max() for c in ids])
    print("    per-CU last workgroup end: min %.1f  median %.1f  max %.1f us  (CUs idle before the launch ends: mean %.1f %% of the span)"
          % (last_end.min(), np.median(last_end), last_end.max(), 100 * np.mean(span - last_end) / span))
    first_gen = starts[: min(n, 256 * 7)]
    print("    first %d starts within %.2f us; start times percentiles (us): %s" % (len(first_gen), first_gen.max(),
          " ".join("%.0f" % np.percentile(t[:, 0], q) for q in (0, 10, 25, 50, 75, 90, 100))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="L1,L2,L4,L5,L7")
    ap.add_argument("--roles", default="fwd,dX")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    lib = _lib.load()
    lib.sdt_debug_set_timeline.argtypes = [ctypes.c_void_p]
    lib.sdt_debug_set_timeline.restype = ctypes.c_int
    B = a.batch
    for name, Hi, Wi, Cin, Cout, kh, kw, s, p in LAYERS:
        if name not in a.only.split(","):
            continue
        one_d = Hi == 1
        x = torch.randn((B, Wi, Cin) if one_d else (B, Hi, Wi, Cin), device="cuda")
        w = torch.nn.Parameter(ops.to_weight_layout(torch.randn((Cout, Cin, kw) if one_d else (Cout, Cin, kh, kw), device="cuda") * 0.05))
        y = ops.conv_forward(x, w, None, s, p)
        gy = torch.randn_like(y)
        flops = 2.0 * y.numel() * Cin * kh * kw
        fns = {"fwd": lambda: ops.conv_forward(x, w, None, s, p), "dX": lambda: ops.conv_input_grad(gy, w, x.shape, s, p)}
        for role in a.roles.split(","):
            for _ in range(3):
                fns[role]()
            buf = torch.zeros((1 << 16, 8), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            assert lib.sdt_debug_set_timeline(ctypes.c_void_p(buf.data_ptr())) == 0
            fns[role]()
            torch.cuda.synchronize()
            lib.sdt_debug_set_timeline(ctypes.c_void_p(0))
            analyse(buf.cpu().numpy().astype(np.uint64), name, role, flops)


if __name__ == "__main__":
    main()
