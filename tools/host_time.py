#!/usr/bin/env python
"""Host-side cost of a train step: time to ENQUEUE forward, backward (autograd thread included) and the optimiser, against
the step's wall time.  python tools/host_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import make_pipeline  # noqa: E402
from speechdrivestemplates_amd import ops  # noqa: E402

pipe, cfg = make_pipeline("voice2pose_sdt_bp", bench.N_CLIPS, batch_global=32)
batches = bench.stage_batches(4, 32, 0, torch.device("cuda", 0))


def step(i, acc=None):
    t0 = time.perf_counter()
    ops.begin_step()
    losses, results = pipe.model(batches[i % 4], pipe.train_dataset)
    t1 = time.perf_counter()
    for o in pipe.optimizers.values():
        o.zero_grad()
    losses["G_loss"].backward()
    t2 = time.perf_counter()
    pipe.optimizer_updates(losses)
    t3 = time.perf_counter()
    if acc is not None:
        acc[0] += t1 - t0
        acc[1] += t2 - t1
        acc[2] += t3 - t2


for i in range(5):
    step(i)
torch.cuda.synchronize()
acc = [0.0, 0.0, 0.0]
n = 20
t0 = time.perf_counter()
for i in range(n):
    step(i, acc)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue: forward %.2f + backward %.2f + optimiser %.2f = %.2f ms/step; wall %.2f ms/step"
      % (acc[0] / n * 1e3, acc[1] / n * 1e3, acc[2] / n * 1e3, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
