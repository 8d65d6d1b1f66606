"""Times the reference's own CPU train step beside the oracle's (SURVEY.md §8d "CPU reference timing").

Runs ONLY in the build container (needs /root/reference, imported under tests/golden/make_golden.py's shim); nothing
here travels to the GPU box or is used by bench.py.  The point is to show that the restatement bench.py times as
``cpu_baseline`` (kind "port") is neither slower nor faster than the original, so the reported GPU/CPU ratio is fair.

    python tests/tools/time_reference_cpu.py [--batch 8] [--steps 3] [--threads 8]
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--clips", type=int, default=64)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)

    import make_golden as MG
    from oracle import sdt_oracle as O
    MG.install_shim()
    from core.pipelines.voice2pose import Voice2PoseModel

    cfg = O.cfg_named("voice2pose_sdt_bp")
    st0 = O.make_voice2pose_state(cfg, a.clips, seed=0, code_std=0.5)
    batches = [O.make_batch(a.batch, a.clips, step=s, seed=1) for s in range(a.steps + 1)]

    # --- the reference: its model + the body of Voice2Pose.train_step (voice2pose.py:281-309) ---------------------
    model = Voice2PoseModel(cfg, None, a.clips)
    model.load_state_dict(MG.clone_state(st0), strict=True)
    model.train()
    ds = MG.ref_dataset(cfg.DATASET.HIERARCHICAL_POSE)
    optG = torch.optim.Adam(model.netG.parameters(), lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    optC = torch.optim.Adam([model.clips_code], lr=cfg.TRAIN.LR)

    def ref_step(batch):
        losses, results = model(batch, ds)
        fin_p = ds.get_final_results(results["poses_pred_batch"].detach(), batch["speaker_stat"])
        fin_g = ds.get_final_results(results["poses_gt_batch"].detach(), batch["speaker_stat"])
        O.evaluate_step(fin_p, fin_g)
        optC.zero_grad()
        optG.zero_grad()
        losses["G_loss"].backward(retain_graph=True)
        optC.step()
        optG.step()
        return float(losses["G_loss"])

    orc = O.OracleVoice2Pose(cfg, MG.clone_state(st0))

    def orc_step(batch):
        return float(orc.train_step(batch)[0]["G_loss"])

    out = {}
    for name, fn in (("reference", ref_step), ("oracle", orc_step)):
        fn(batches[0])  # warm-up
        t0 = time.perf_counter()
        last = [fn(b) for b in batches[1:]]
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = (a.batch / dt, last[-1])
        print("%-9s %.3f s/step  %.2f clips/s  (G_loss after %d steps %.6f)" % (name, dt, a.batch / dt, a.steps + 1, last[-1]))
    print("oracle / reference speed ratio: %.3f  (threads=%d, batch=%d)" % (out["oracle"][0] / out["reference"][0], a.threads, a.batch))


if __name__ == "__main__":
    main()
