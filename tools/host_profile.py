#!/usr/bin/env python
"""cProfile of the host side of the train step (where do the ~6 ms of Python per step go?).  python tools/host_profile.py [config]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import make_pipeline  # noqa: E402

CFG = sys.argv[1] if len(sys.argv) > 1 else "voice2pose_sdt_bp"
pipe, cfg = make_pipeline(CFG, bench.N_CLIPS, batch_global=32)
batches = bench.stage_batches(4, 32, 0, torch.device("cuda", 0))


def step(i):
    losses, _ = pipe.forward_backward(batches[i % 4])
    pipe.optimizer_updates(losses)


for i in range(5):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
