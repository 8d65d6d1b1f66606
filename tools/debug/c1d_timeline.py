#!/usr/bin/env python
"""Per-step timeline of the fused Conv1d kernel (csrc/conv1d.hip, -DSDT_TUNING build): s_memrealtime stamps of workgroup 0's
first loader and first multiplier wave for every launch of one train step.
    python __graft_entry__.py --tuning && python tools/debug/c1d_timeline.py"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["SDT_HIP_LIB"] = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import make_pipeline  # noqa: E402
from speechdrivestemplates_amd import _lib  # noqa: E402

pipe, cfg = make_pipeline("voice2pose_sdt_bp", bench.N_CLIPS, batch_global=32)
batches = bench.stage_batches(2, 32, 0, torch.device("cuda", 0))
for i in range(3):
    losses, _ = pipe.forward_backward(batches[i % 2])
    pipe.optimizer_updates(losses)
torch.cuda.synchronize()
lib = _lib.load()
N = 40
buf = torch.zeros((N, 64), dtype=torch.int64, device="cuda")
lib.sdt_c1d_debug_buffer.argtypes = [C.c_void_p, C.c_int]
lib.sdt_c1d_debug_buffer.restype = None
lib.sdt_c1d_debug_buffer(buf.data_ptr(), N)
lib.sdt_c1d_debug_mode.argtypes = [C.c_int]
lib.sdt_c1d_debug_mode.restype = None
lib.sdt_c1d_debug_mode(int(os.environ.get("C1D_DBG_MODE", "0")))
losses, _ = pipe.forward_backward(batches[0])
torch.cuda.synchronize()
lib.sdt_c1d_debug_buffer(None, 0)
lib.sdt_c1d_debug_mode(0)
ONLY = [int(x) for x in os.environ.get("C1D_LAUNCHES", "0,2,6,8,14,18,27").split(",")]
t = buf.cpu().numpy()
tick_us = 0.01  # s_memrealtime: 100 MHz
for i in range(N):
    L, Mw = t[i, :32], t[i, 32:]
    if L[0] == 0 or i not in ONLY:
        continue
    t0 = min(L[0], Mw[0])
    def rel(a):
        return " ".join("%5.1f" % ((x - t0) * tick_us) if x else "    -" for x in a)
    print("launch %2d loader: start %s | issue %s tables %s b1 %s lds0 %s b2 %s | steps %s | end %s %s" % (
        i, rel(L[0:1]), rel(L[1:2]), rel(L[2:3]), rel(L[3:4]), rel(L[4:5]), rel(L[5:6]), rel(L[6:30]), rel(L[30:31]), rel(L[31:32])))
    print("          mult  : start %s | pre %s b1 %s b2 %s | steps %s | end %s %s" % (
        rel(Mw[0:1]), rel(Mw[2:3]), rel(Mw[3:4]), rel(Mw[5:6]), rel(Mw[6:30]), rel(Mw[30:31]), rel(Mw[31:32])))
