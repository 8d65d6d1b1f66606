#!/usr/bin/env python
"""What does a collective's long-lived kernel do to the persistent conv kernels?  (Only 1-GPU boxes are available: the data-parallel exchange is
emulated by `wgs` spinning workgroups of `us` microseconds on a third stream, launched where the first gradient bucket's all-reduce would be
launched -- when backward reaches the audio encoder -- plus a second one half-way through the Conv2d backward.)
    python __graft_entry__.py --tuning
    python tools/debug/comm_emulation.py [--wgs 32] [--us 1200] [--grid 512|480|448]"""
import argparse
import ctypes
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["SDT_HIP_LIB"] = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import make_pipeline  # noqa: E402
from speechdrivestemplates_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wgs", type=int, default=32)
    ap.add_argument("--us", type=int, default=1200)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reserve", type=int, default=0, help="workgroup slots the BACKWARD stream-K plans leave free (ops.SK_RESERVED_SLOTS -> sdt_convsk_set_reserved_slots)")
    ap.add_argument("--lds", type=int, default=65536, help="LDS bytes per spinning workgroup (65536: cannot share a CU with two conv workgroups)")
    ap.add_argument("--no-chain1d", action="store_true", help="the generator's Conv1d stage block by block (A/B of the persistent chain launch under a collective)")
    a = ap.parse_args()
    ops.CHAIN1D = not a.no_chain1d
    lib = _lib.load()
    lib.sdt_debug_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.sdt_debug_spin.restype = ctypes.c_int
    ops.SK_RESERVED_SLOTS = a.reserve  # what dp.GradReducer sets in a data-parallel run: backward plans leave this many slots free
    pipe, _ = make_pipeline("voice2pose_sdt_bp", bench.N_CLIPS, batch_global=32, sys_opts={"CHAIN1D": not a.no_chain1d})
    batches = bench.stage_batches(4, 32, 0, torch.device("cuda", 0))
    comm = torch.cuda.Stream()
    state = {"on": False}
    orig = ops.flush_deferred_dw

    def hooked():
        pending = bool(ops._DEFERRED)  # the flush from the post-encoder hook (the step's closing flush in forward_backward finds nothing queued:
        orig()                         # a collective launched THERE would start behind the whole backward pass and run serially)
        if state["on"] and pending:
            comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm):
                _lib.check(lib.sdt_debug_spin(a.wgs, a.us, a.lds, ctypes.c_void_p(comm.cuda_stream)))
    ops.flush_deferred_dw = hooked
    import speechdrivestemplates_amd.core.networks.keypoints_generation.generator as gen
    gen.ops.flush_deferred_dw = hooked

    def run(n):
        for i in range(5):
            losses, _ = pipe.forward_backward(batches[i % 4])
            pipe.optimizer_updates(losses)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            losses, _ = pipe.forward_backward(batches[i % 4])
            torch.cuda.current_stream().wait_stream(comm)
            pipe.optimizer_updates(losses)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    for on in (False, True, False, True):
        state["on"] = on
        ms = run(a.steps)
        print("reserve %3d  emulated collective %-3s (%d workgroups x %d us per step, %d KB of LDS each): %.3f ms/step  %.0f clips/s" % (a.reserve, "on" if on else "off", a.wgs, a.us, a.lds // 1024, ms, 32 / ms * 1e3))
    assert not ops.streamk_error_codes(), ops.streamk_error_codes()


if __name__ == "__main__":
    main()
