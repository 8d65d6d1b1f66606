"""Edge batches through the whole train step against the CPU oracle: B = 1 (BatchNorm / KL with one sample per batch where the config has them),
B = 3, 7 (ragged tiles everywhere), B = 33 (one clip more than the plans are tuned for) -- step-0 losses and metrics (tight), headline config.
python tools/debug/odd_batches.py      (test infrastructure: imports oracle/)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from __graft_entry__ import make_pipeline  # noqa: E402
from oracle import sdt_oracle as O  # noqa: E402


def run(cfg_name, B, n_clips=40):
    ocfg = O.cfg_named(cfg_name)
    state = O.make_voice2pose_state(ocfg, n_clips, seed=0, code_std=0.5)
    pipe, _ = make_pipeline(cfg_name, n_clips, state={k: v.clone() for k, v in state.items()})
    eng = O.OracleVoice2Pose(ocfg, state)
    batch = O.make_batch(B, n_clips, step=0, seed=5)
    losses, results = pipe.forward_backward(batch)
    pipe.optimizer_updates(losses)
    ref_losses, ref_results = eng.train_step(batch)
    torch.cuda.synchronize()
    worst = 0.0
    for k, v in ref_losses.items():
        if k in losses and torch.is_tensor(v) and v.numel() == 1:
            a, b = float(losses[k]), float(v)
            if b != b and a != a:
                continue  # (B = 1: an unbiased variance of one sample is NaN in the reference too)
            rel = abs(a - b) / (abs(b) + 1e-6)
            worst = max(worst, rel)
            assert rel < 5e-5, (cfg_name, B, k, a, b)
    from speechdrivestemplates_amd import ops
    assert not ops.streamk_error_codes()
    print("%-20s B=%2d  losses within %.1e of the oracle  (%s)" % (cfg_name, B, worst, ", ".join("%s=%.5f" % (k, float(v)) for k, v in list(losses.items())[:3])), flush=True)


if __name__ == "__main__":
    for cfg in ("voice2pose_sdt_bp",):  # (s2g needs the global speaker statistics handed to the oracle, pose2pose its noise: tests/test_model_gpu.py does both at B = 4)
        for B in (1, 3, 7, 33):
            try:
                run(cfg, B)
            except Exception as e:  # report and go on: this is a survey
                import traceback
                print("%-20s B=%2d  FAILED: %r\n%s" % (cfg, B, e, "".join(traceback.format_exc().splitlines(True)[-6:])), flush=True)
