"""Debug / evidence: at B=32 the per-tensor distance of a gradient from float64 is dominated by discrete branch events
(LeakyReLU / |.| kinks crossed by 1e-6 rounding differences), for the HIP path and for the fp32 oracle alike: which tensors are
hit, and how hard, changes with the batch, in both implementations."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import test_fullsize_gpu as T
from oracle import sdt_oracle as O

orig = O.make_batch
for seed in (11, 12, 13, 14):
    O.make_batch = lambda B, N, step=0, seed=1, _s=seed, **kw: orig(B, N, step=step, seed=_s, **kw)
    r = T._b32_run("voice2pose_sdt_bp", 0.5)
    names = sorted(r["g64_hip"])
    eh = {k: T._relmax(r["grads_hip"][k], r["g64_hip"][k]) for k in names}
    er = {k: T._relmax(r["g32"][k], r["g64_ref"][k]) for k in names}
    top_h = sorted(names, key=lambda k: -eh[k])[:3]
    top_r = sorted(names, key=lambda k: -er[k])[:3]
    print("seed %d: median hip %.2e ref32 %.2e | max hip %.2e ref32 %.2e | L2 hip %.2e ref32 %.2e" % (
        seed, np.median(list(eh.values())), np.median(list(er.values())), max(eh.values()), max(er.values()),
        T._flat_l2(r["grads_hip"], r["g64_hip"], names), T._flat_l2(r["g32"], r["g64_ref"], names)))
    print("   worst hip  :", [(k.replace("netG.", "").replace("audio_encoder.specgram_encoder_2d", "enc2d"), "%.1e" % eh[k], "ref %.1e" % er[k]) for k in top_h])
    print("   worst ref32:", [(k.replace("netG.", "").replace("audio_encoder.specgram_encoder_2d", "enc2d"), "%.1e" % er[k], "hip %.1e" % eh[k]) for k in top_r])
