#!/usr/bin/env python
"""bf16 storage at MODEL level against the bf16-emulating oracle: where along the generator does the residual enter?"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import sdt_oracle as O  # noqa: E402
from speechdrivestemplates_amd import ops  # noqa: E402
from test_model_gpu import _make_pipeline  # noqa: E402


def rms(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def main():
    B, N, cfg_name = 8, 64, "voice2pose_sdt_bp"
    ocfg = O.cfg_named(cfg_name)
    state = O.make_voice2pose_state(ocfg, N, seed=0, code_std=0.5)
    batch = O.make_batch(B, N, step=3, seed=11)
    st64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
    mel64 = O.mel_spectrogram(batch["audio"].double(), st64.get("mel_transfm.spectrogram.window"), st64.get("mel_transfm.mel_scale.fb"))
    ops.set_storage("bf16")
    ops.CHAIN_MATH = "f32"
    pipe, _ = _make_pipeline(cfg_name, N, 0.5)
    net = pipe.model.netG
    dev = torch.device("cuda", 0)
    with torch.no_grad():
        mel_hip = pipe.model.mel_transfm(batch["audio"].to(dev)) if hasattr(pipe.model, "mel_transfm") else None
        print("mel: hip vs float64 rms %.3e" % rms(mel_hip, mel64))
        # engine blocks one by one on the HIP mel
        x = mel_hip.unsqueeze(-1)
        outs = []
        i = 0
        for stage in net.audio_encoder.specgram_encoder_2d:
            for block in stage:
                x = block.forward_cl(x, None, None, out_f32=(i == 7))
                outs.append(x)
                i += 1
        # oracle blocks, plain and emulated, on the float64 mel
        for emu_on in (False, True):
            xo = mel64.unsqueeze(1)
            last = len(O.AUDIO_ENCODER_2D) - 1
            for i, (_, _, _, s, p) in enumerate(O.AUDIO_ENCODER_2D):
                emu = None if not emu_on else ("l0" if i == 0 else ("2d_last" if i == last else "2d"))
                xo = O.conv_norm_act(xo, st64, "netG.audio_encoder.specgram_encoder_2d.%d.%d" % (i // 2, i % 2), s, p, "IN", True, True, emu)
                print("  block %d  %s oracle: rms(hip - oracle) %.3e" % (i, "emulating" if emu_on else "plain    ", rms(outs[i].float().permute(0, 3, 1, 2), xo)))
    ops.set_storage("f32")
    ops.CHAIN_MATH = None


if __name__ == "__main__":
    main()
