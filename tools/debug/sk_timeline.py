#!/usr/bin/env python
"""Per-segment timeline of the persistent stream-K conv (tuning build): where a workgroup's time goes between tiles.
   python __graft_entry__.py --tuning && python tools/debug/sk_timeline.py [--only L2,L5] [--roles fwd,dX] [--wpc 2]"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["SDT_HIP_LIB"] = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speechdrivestemplates_amd import _lib, ops  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tools"))
from conv_bench import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="L1,L2,L4,L5")
    ap.add_argument("--roles", default="fwd,dX")
    ap.add_argument("--wpc", type=int, default=2)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="bf16: the bf16-storage kernels (forward with statistics, dX)")
    ap.add_argument("--bf2", action="store_true", help="with --dtype bf16: the bf16-shaped kernel of convbf.hip (256-row tiles, one workgroup per CU) -- "
                    "stamps: segment start, fill done, K loop done, next tile's set-up + first request issued, end phase done")
    a = ap.parse_args()
    ops.BF16_SHAPED = bool(a.bf2)
    ops.F32_SPLIT = bool(a.bf2) and a.dtype == "f32"  # --dtype f32 --bf2: the split-fp32 form of the 8-wave kernel
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    lib = _lib.load()
    set_tl = lib.sdt_debug_set_timeline_bf2 if a.bf2 else lib.sdt_debug_set_timeline_sk
    set_tl.argtypes = [ctypes.c_void_p]
    if not a.bf2:
        _lib.check(lib.sdt_convsk_set_wg_per_cu(a.wpc))
    B = 32
    for name, Hi, Wi, Cin, Cout, kh, kw, s, p in LAYERS:
        if name not in a.only.split(",") or Hi == 1:
            continue
        x = torch.randn((B, Hi, Wi, Cin), device="cuda").to(dt)
        w = torch.nn.Parameter(ops.to_weight_layout(torch.randn((Cout, Cin, kh, kw), device="cuda") * 0.05))
        if a.dtype == "bf16":
            y = ops.ConvStatsFn.apply(x, w, s, p, B, None)[0]

            def fwd():
                ops._ARENA.begin_step(torch.device("cuda", 0))
                return ops.ConvStatsFn.apply(x, w, s, p, B, None)
        else:
            y = ops.conv_forward(x, w, None, s, p)

            def fwd():
                return ops.conv_forward(x, w, None, s, p)
        gy = torch.randn_like(y)
        fns = {"fwd": fwd, "dX": lambda: ops.conv_input_grad(gy, w, x.shape, s, p)}
        for role in a.roles.split(","):
            for _ in range(3):
                fns[role]()
            buf = torch.zeros((512 * 16, 8), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            assert set_tl(ctypes.c_void_p(buf.data_ptr())) == 0
            fns[role]()
            torch.cuda.synchronize()
            set_tl(ctypes.c_void_p(0))
            tl = buf.cpu().numpy().astype(np.int64).reshape(512, 16, 8)
            np.save(os.path.join(REPO, "gpurun_out", "sk_tl_%s_%s.npy" % (name, role)), tl)
            used = tl[:, :, 0] != 0
            t = tl[..., :5].astype(np.float64) * 0.01
            t0 = t[..., 0][used].min()
            span = t[..., 4][used].max() - t0
            steps = tl[..., 5][used]
            kind = tl[..., 6][used]
            if a.bf2:
                ph = {"fill": (t[..., 1] - t[..., 0])[used], "K loop": (t[..., 2] - t[..., 1])[used], "next tile's set-up + first request": (t[..., 3] - t[..., 2])[used],
                      "end (publish / combine / epilogue)": (t[..., 4] - t[..., 3])[used]}
            else:
                ph = {"set-up": (t[..., 1] - t[..., 0])[used], "fill": (t[..., 2] - t[..., 1])[used], "K loop": (t[..., 3] - t[..., 2])[used],
                      "end (publish / combine / epilogue)": (t[..., 4] - t[..., 3])[used]}
            nseg = used.sum(1)
            print("%s %s: span %.1f us; segments per workgroup %.1f (max %d, 16 recorded at most); K steps per segment median %d; kinds whole/owner/publish %d/%d/%d"
                  % (name, role, span, nseg[nseg > 0].mean(), nseg.max(), np.median(steps), (kind == 0).sum(), (kind == 1).sum(), (kind == 2).sum()))
            for k, v in ph.items():
                print("    %-36s median %6.2f us  p90 %6.2f  max %6.2f   sum per workgroup %.1f us" % (k, np.median(v), np.percentile(v, 90), v.max(), v.sum() / (nseg > 0).sum()))
            per_step = ph["K loop"] / np.maximum(steps, 1)
            print("    K loop per step: median %.3f us = %.0f cycles at 2.38 GHz (%d MFMA cycles per SIMD at 32 per MFMA, 256 x 128 tile; half of that for 128 x 128 and 256 x 64)" % (np.median(per_step), np.median(per_step) * 2380,
                                                                                                         (1024 if a.dtype == "bf16" else 3072) if a.bf2 else (512 if a.dtype == "bf16" else 4096)))
            if a.bf2 and a.dtype == "f32":
                bw = tl[..., 7][used].astype(np.float64) / 8.0 / np.maximum(steps, 1)  # clock64 ticks per step and wave
                print("    barrier wait per K step and wave: median %.0f clock64 ticks (p90 %.0f) = %.0f %% of a step at the shader clock"
                      % (np.median(bw), np.percentile(bw, 90), 100.0 * np.median(bw) / (np.median(per_step) * 1950.0)))
            for kk, nm in ((0, "whole"), (1, "owner"), (2, "publish")):
                sel = kind == kk
                if sel.any():
                    print("    end phase of %-8s segments: median %.2f us  p90 %.2f" % (nm, np.median(ph["end (publish / combine / epilogue)"][sel]), np.percentile(ph["end (publish / combine / epilogue)"][sel], 90)))
            first = np.array([t[i, :, 0][used[i]].min() for i in range(512) if used[i].any()]) - t0
            print("    workgroup start times: median %.1f max %.1f us" % (np.median(first), first.max()))
            last = np.array([t[i, :, 4][used[i]].max() for i in range(512) if used[i].any()]) - t0
            print("    workgroup finish times: min %.1f median %.1f max %.1f us" % (last.min(), np.median(last), last.max()))


if __name__ == "__main__":
    main()
