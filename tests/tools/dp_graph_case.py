"""One data-parallel case in its OWN process (tests/test_dp_gpu.py spawns it): a 1-rank RCCL process group with SDT_DP_FORCE=1 -- the whole
exchange path of dp.GradReducer over RCCL on the one GPU a test box has -- stepping a pipeline eagerly or through graph.GraphedStep
("full": the all-reduces captured with the step; "split": graph segments around an eager exchange).  Writes losses + weights to --out (npz).
A process of its own because a failed stream capture leaves a sticky HIP error behind that would take the rest of the pytest session with it."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="voice2pose_sdt_bp")
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--mode", default="eager", choices=["eager", "full", "split"])
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--no-dp", action="store_true", help="plain single-process run (no process group): the reference trajectory")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    if not a.no_dp:
        os.environ["SDT_DP_FORCE"] = "1"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % a.port, rank=0, world_size=1)
    from oracle import sdt_oracle as O
    from speechdrivestemplates_amd import ops
    from speechdrivestemplates_amd.graph import GraphedStep
    from test_model_gpu import _make_pipeline
    ops.set_storage(a.storage)
    torch.manual_seed(3)
    code_std = 0.5 if a.config == "voice2pose_sdt_bp" else 0.0
    pipe, _ = _make_pipeline(a.config, 16, code_std)
    assert pipe.reducer.active == (not a.no_dp)
    dev = torch.device("cuda", 0)
    gs = GraphedStep(pipe, warmup=1, mode=None if a.mode == "eager" else a.mode)
    key = "G_loss" if a.config.startswith("voice2pose") else "loss"
    hist = []
    for step in range(a.steps):
        b = O.make_batch(a.batch, 16, step=step, seed=1)
        b = {k: (v.to(dev) if torch.is_tensor(v) and k != "num_frames" else v) for k, v in b.items()}
        b["speaker_stat"] = {k: v.to(dev) for k, v in b["speaker_stat"].items()}
        if a.config == "voice2pose_s2g":
            b["speaker"] = ["oliver"] * a.batch
        if a.mode == "eager":
            losses, _ = pipe.forward_backward(b)
            pipe.optimizer_updates(losses)
        else:
            losses = gs.run(b)
        torch.cuda.synchronize()
        hist.append([float(losses[key]), float(losses["L2_dist"])])
    assert not ops.streamk_error_codes()
    opt = next(iter(pipe.optimizers.values()))
    segs = None if gs.segments is None else [k for k, _ in gs.segments]
    np.savez(a.out, hist=np.asarray(hist), flat=opt.flat_param.detach().cpu().numpy(), steps=int(opt.state_dev[0]))
    print(json.dumps({"hist": hist, "segments": segs, "mode": gs.mode, "reserve": ops.SK_RESERVED_SLOTS}))
    if not a.no_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
