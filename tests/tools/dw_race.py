"""Debug: per-tensor run-to-run spread of the weight gradients at B=32 with the side-stream paths on / off."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from oracle import sdt_oracle as O
from test_model_gpu import _make_pipeline
from speechdrivestemplates_amd import ops

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "voice2pose_sdt_vae"
batch = O.make_batch(32, 64, step=3, seed=11)
runs = {}
for tag, ov, de, aux in (("inline", False, False, False), ("inline2", False, False, False), ("aux", False, False, True), ("overlap", True, False, False),
                         ("overlap+defer", True, True, False), ("all", True, True, True), ("all2", True, True, True)):
    ops.OVERLAP_DW, ops.DEFER_SMALL_DW, ops.OVERLAP_AUX = ov, de, aux
    pipe, _ = _make_pipeline(cfg_name, 64, 0.5 if cfg_name.endswith("bp") else 0.0)
    losses, res = pipe.forward_backward(batch)
    torch.cuda.synchronize()
    runs[tag] = {k: p.grad.detach().double().cpu() for k, p in pipe.model.named_parameters() if p.grad is not None}
base = runs["inline"]
for tag, g in runs.items():
    bad = []
    for k in base:
        e = ((g[k] - base[k]).abs().max() / base[k].abs().max().clamp_min(1e-30)).item()
        if e > 2e-5:
            bad.append((k.replace("netG.", ""), "%.1e" % e))
    print("%-14s tensors deviating > 2e-5 from 'inline': %d %s" % (tag, len(bad), bad[:12]))
