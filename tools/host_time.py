import sys, time, torch
sys.path.insert(0, ".")
import bench
from __graft_entry__ import make_pipeline
pipe, cfg = make_pipeline("voice2pose_sdt_bp", bench.N_CLIPS, batch_global=32)
batches = bench.stage_batches(4, 32, 0, torch.device("cuda", 0))
def step(i):
    losses, _ = pipe.forward_backward(batches[i % 4]); pipe.optimizer_updates(losses)
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
