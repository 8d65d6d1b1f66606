#!/usr/bin/env python
"""voice2pose_s2g B=4 step-0 gradients with the stream-K kernels on / off, per parameter; logs every plan's geometry."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ["SDT_HIP_LIB"] = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
import torch  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402
from oracle import sdt_oracle as O  # noqa: E402
from test_model_gpu import _make_pipeline  # noqa: E402

res = {}
orig = ops._sk_plan


def logged(garr, n, rpg, bwd_groups, dev, forward=False):
    plan = orig(garr, n, rpg, bwd_groups, dev, forward)
    g = garr if isinstance(garr, ops.ConvGeom) else garr[0]
    key = (g.B, g.Hi, g.Wi, g.Cin, g.Cout, g.ntaps, g.Ho, g.Wo, n, rpg, bwd_groups, forward)
    if key not in logged.seen:
        logged.seen.add(key)
        print("plan", key, "->", None if plan is None else plan.kind)
    return plan


logged.seen = set()
ops._sk_plan = logged
import ctypes  # noqa: E402
from speechdrivestemplates_amd import _lib  # noqa: E402
lib = _lib.load()
for sk in (False, True):
    lib.sdt_convsk_set_perm_pct(ctypes.c_int(int(os.environ.get("PCT_B" if sk else "PCT_A", "95" if sk else "0"))))  # tuning library: A/B of the image-row-major tile order
    ops._SK_PLANS.clear()
    pipe, cfg = _make_pipeline("voice2pose_s2g", 16, 0.0)
    batch = O.make_batch(4, 16, step=0, seed=1)
    batch["speaker"] = ["oliver"] * 4
    losses, results = pipe.forward_backward(batch)
    torch.cuda.synchronize()
    res[sk] = {k: p.grad.detach().clone() for k, p in pipe.model.named_parameters() if p.grad is not None}
    res[(sk, "pred")] = results["poses_pred_normalized"].detach().clone()
    print("perm" if sk else "natural", "pred absmax", res[(sk, "pred")].abs().max().item())
print("pred diff", (res[(True, "pred")] - res[(False, "pred")]).abs().max().item())
for k in res[True]:
    a, b = res[True][k].double(), res[False][k].double()
    e = ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    if e > 1e-4:
        print("%-50s rel diff %.2e" % (k, e))
