#!/usr/bin/env python
"""Host-side cost of a train step: the time the host needs to ENQUEUE forward, backward (autograd thread included) and the optimiser,
against the step's wall time.  The enqueue times are taken over a burst that never waits for the GPU (ops.MAX_STEPS_IN_FLIGHT = 0:
begin_step's fence would otherwise charge the GPU's lag to `forward`).  python tools/host_time.py [--storage bf16] [--graph]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import make_pipeline  # noqa: E402
from speechdrivestemplates_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--storage", default="f32", choices=["f32", "bf16"])
ap.add_argument("--config", default="voice2pose_sdt_bp")
ap.add_argument("--graph", action="store_true", help="also time hipGraph replay of the whole step")
args = ap.parse_args()
ops.set_storage(args.storage)
pipe, cfg = make_pipeline(args.config, bench.N_CLIPS, batch_global=32, sys_opts={"STORAGE": args.storage})
batches = bench.stage_batches(4, 32, 0, torch.device("cuda", 0))


def full_step(i, acc=None):
    t0 = time.perf_counter()
    losses, _ = pipe.forward_backward(batches[i % 4])
    t1 = time.perf_counter()
    pipe.optimizer_updates(losses)
    t2 = time.perf_counter()
    if acc is not None:
        acc[0] += t1 - t0
        acc[1] += t2 - t1


for i in range(6):
    full_step(i)
torch.cuda.synchronize()
# (a) pure enqueue cost: a burst the host never waits in
lead = ops.MAX_STEPS_IN_FLIGHT
ops.MAX_STEPS_IN_FLIGHT = 0
acc, n = [0.0, 0.0], 8
t0 = time.perf_counter()
for i in range(n):
    full_step(i, acc)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
ops.MAX_STEPS_IN_FLIGHT = lead
# (b) steady state wall time
torch.cuda.synchronize()
t0 = time.perf_counter()
m = 30
for i in range(m):
    full_step(i)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / m
print("storage %s: host enqueue forward+backward %.2f + optimiser %.2f = %.2f ms/step (burst of %d: %.2f ms/step until the last launch was queued, "
      "%.2f ms/step until the GPU finished); steady-state wall %.2f ms/step -> %s-bound"
      % (args.storage, 1e3 * acc[0] / n, 1e3 * acc[1] / n, 1e3 * sum(acc) / n, n, 1e3 * t_host / n, 1e3 * t_all / n, 1e3 * wall,
         "host" if t_host / n > 0.9 * wall else "GPU"))
if args.graph:
    from speechdrivestemplates_amd.graph import GraphedStep
    gs = GraphedStep(pipe, warmup=1)
    for i in range(4):
        gs.run(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(m):
        gs.run(batches[i % 4])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tg = time.perf_counter() - t0
    print("hipGraph replay: host %.2f ms/step, wall %.2f ms/step" % (1e3 * th / m, 1e3 * tg / m))
assert not ops.streamk_error_codes()
