"""CPU oracle for the SDT voice2pose training hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; nothing under ``speechdrivestemplates_amd/`` does.  It is a *restatement* of the
reference algorithm in PyTorch-CPU (any float dtype), written functionally over a
flat ``state`` dict whose keys are the reference's ``state_dict`` names, so that a
reference checkpoint, the oracle and the HIP path can all be driven from the same
tensors.  Every function cites the reference file:line it follows (paths are
relative to the upstream repo ShenhanQian/SpeechDrivesTemplates).

Pinning status (see tests/golden/make_golden.py and DESIGN.md):
  * networks, losses, metrics, train-step trajectory: pinned against outputs of the
    reference's own modules imported in the authoring container (fixtures under
    tests/golden/, checked by tests/test_oracle_golden.py).
  * mel front end: the reference calls torchaudio==0.7.0 (requirements.txt:9), which
    is neither vendored in the reference nor installed here.  ``mel_spectrogram``
    restates torchaudio 0.7.0's documented algorithm on top of ``torch.stft`` and is
    cross-checked against an independent float64 DFT -- **mel parity unpinned**.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Layer tables.  (cin, cout, kernel, stride, pad); "down" = k4 s2 p1, "same" = k3 s1 p1
# building_blocks.py:8-12 fixes those two shapes; generator.py:29 adds the (6,3) p0 one.
# --------------------------------------------------------------------------------------
AUDIO_ENCODER_2D = [  # generator.py:15-30
    (1, 64, (3, 3), 1, 1), (64, 64, (4, 4), 2, 1),
    (64, 128, (3, 3), 1, 1), (128, 128, (4, 4), 2, 1),
    (128, 256, (3, 3), 1, 1), (256, 256, (4, 4), 2, 1),
    (256, 256, (3, 3), 1, 1), (256, 256, (6, 3), 1, 0),
]
UNET_ENC = [("e0", False), ("e1", False), ("e2", True), ("e3", True),
            ("e4", True), ("e5", True), ("e6", True)]  # generator.py:53-62
UNET_DEC = ["d5", "d4", "d3", "d2", "d1"]  # generator.py:64-68
POSE_ENC_DOWN = [False, False, True, True, True, True, True]  # autoencoder.py:17-25
DISC_LAYERS = [(None, 256, 4, 2, 1), (256, 512, 4, 2, 1), (512, 1024, 3, 1, 1)]  # discriminator.py:12-16

MEL_N_FFT, MEL_WIN, MEL_HOP, MEL_FMIN, MEL_FMAX, MEL_NMELS, MEL_SR = 512, 400, 160, 55.0, 7500.0, 80, 16000


def default_cfg(**over):
    """Attribute-access config with the effective defaults of configs/default.py:4-97
    for the keys the networks / models read.  ``over`` uses dotted keys."""
    cfg = SimpleNamespace(
        VOICE2POSE=SimpleNamespace(
            GENERATOR=SimpleNamespace(
                NAME="SequenceGeneratorCNN", LEAKY_RELU=True, NORM="IN", LAMBDA_REG=1.0, LAMBDA_CLIP_KL=0.1,
                CLIP_CODE=SimpleNamespace(DIMENSION=None, LR_SCALING=1.0, TRAIN=True, FRAME_VARIANT=False,
                                          SAMPLE_FROM_NORMAL=False, TEST_WITH_GT_CODE=False,
                                          EXTERNAL_CODE=False, EXTERNAL_CODE_PTH=None)),
            POSE_ENCODER=SimpleNamespace(NAME="PoseSeqEncoder", AE_CHECKPOINT=None),
            POSE_DISCRIMINATOR=SimpleNamespace(NAME=None, LEAKY_RELU=False, LAMBDA_GAN=1.0, MOTION=True,
                                               WHITE_LIST=None),
            STRICT_LOADING=True),
        POSE2POSE=SimpleNamespace(
            AUTOENCODER=SimpleNamespace(NAME=None, LEAKY_RELU=True, NORM="BN", CODE_DIM=32),
            LAMBDA_REG=1.0, LAMBDA_KL=0.1),
        DATASET=SimpleNamespace(NUM_LANDMARKS=121, HIERARCHICAL_POSE=True, NUM_FRAMES=64, AUDIO_LENGTH=68267,
                                AUDIO_SR=16000, FPS=15),
        TRAIN=SimpleNamespace(LR=1e-4, WD=0, BATCH_SIZE=32),
        DEMO=SimpleNamespace(CODE_INDEX=None, CODE_INDEX_B=None),
    )
    for k, v in over.items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = getattr(node, p)
        setattr(node, parts[-1], v)
    return cfg


def cfg_named(name):
    """The four BASELINE configs (configs/*.yaml over configs/default.py)."""
    if name == "voice2pose_s2g":  # configs/voice2pose_s2g.yaml
        return default_cfg(**{"VOICE2POSE.GENERATOR.NORM": "BN",
                              "VOICE2POSE.POSE_DISCRIMINATOR.NAME": "PoseSequenceDiscriminator",
                              "VOICE2POSE.POSE_DISCRIMINATOR.LAMBDA_GAN": 0.1,
                              "VOICE2POSE.POSE_DISCRIMINATOR.LEAKY_RELU": True,
                              "DATASET.HIERARCHICAL_POSE": False})
    if name == "voice2pose_sdt_bp":  # configs/voice2pose_sdt_bp.yaml
        return default_cfg(**{"VOICE2POSE.GENERATOR.CLIP_CODE.DIMENSION": 32})
    if name == "voice2pose_sdt_vae":  # configs/voice2pose_sdt_vae.yaml
        return default_cfg(**{"VOICE2POSE.GENERATOR.CLIP_CODE.DIMENSION": 32,
                              "VOICE2POSE.GENERATOR.CLIP_CODE.EXTERNAL_CODE": True})
    if name == "pose2pose":  # configs/pose2pose.yaml
        return default_cfg(**{"POSE2POSE.AUTOENCODER.NAME": "Autoencoder"})
    raise KeyError(name)


# --------------------------------------------------------------------------------------
# Mel front end (torchaudio==0.7.0 MelSpectrogram as configured at voice2pose.py:27-30)
# --------------------------------------------------------------------------------------
def mel_window(dtype=torch.float32):
    """hann_window(400), periodic -- torchaudio 0.7 Spectrogram default window_fn."""
    return torch.hann_window(MEL_WIN, periodic=True, dtype=dtype)


def mel_filterbank(dtype=torch.float32):
    """torchaudio 0.7 functional.create_fb_matrix(257, 55, 7500, 80, 16000, norm=None): HTK mel
    scale, triangular filters, float32 arithmetic as upstream."""
    n_freqs = MEL_N_FFT // 2 + 1
    all_freqs = torch.linspace(0, MEL_SR // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + MEL_FMIN / 700.0)
    m_max = 2595.0 * math.log10(1.0 + MEL_FMAX / 700.0)
    m_pts = torch.linspace(m_min, m_max, MEL_NMELS + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    return fb.to(dtype)  # (257, 80)


def stft_power(audio, window):
    """|STFT|^2 with center=True / reflect pad / onesided (torchaudio 0.7 functional.spectrogram)."""
    spec = torch.stft(audio, MEL_N_FFT, hop_length=MEL_HOP, win_length=MEL_WIN, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return spec.real ** 2 + spec.imag ** 2  # (B, 257, F)


def mel_spectrogram(audio, window=None, fb=None):
    """audio (B, L) -> power mel (B, 80, 1 + L // 160); no log (voice2pose.py:125)."""
    window = mel_window(audio.dtype) if window is None else window
    fb = mel_filterbank(audio.dtype) if fb is None else fb
    power = stft_power(audio, window)
    return torch.matmul(power.transpose(1, 2), fb).transpose(1, 2)


def stft_power_dft64(audio):
    """Independent float64 direct-DFT statement of the same STFT (self-check of the restatement)."""
    a = audio.double()
    pad = MEL_N_FFT // 2
    ap = F.pad(a.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    n_frames = 1 + a.shape[-1] // MEL_HOP
    frames = ap.unfold(-1, MEL_N_FFT, MEL_HOP)[:, :n_frames]  # (B, F, 512)
    w = torch.zeros(MEL_N_FFT, dtype=torch.float64)
    left = (MEL_N_FFT - MEL_WIN) // 2
    w[left:left + MEL_WIN] = torch.hann_window(MEL_WIN, periodic=True, dtype=torch.float64)
    n = torch.arange(MEL_N_FFT, dtype=torch.float64)
    k = torch.arange(MEL_N_FFT // 2 + 1, dtype=torch.float64)
    ang = 2 * math.pi * torch.outer(n, k) / MEL_N_FFT
    fw = frames * w
    re = fw @ torch.cos(ang)
    im = -(fw @ torch.sin(ang))
    return (re ** 2 + im ** 2).transpose(1, 2)


# --------------------------------------------------------------------------------------
# ConvNormRelu (building_blocks.py:4-55)
# --------------------------------------------------------------------------------------
def _norm(y, state, prefix, norm, training, momentum=0.1, eps=1e-5):
    if norm == "IN":
        if y.dim() == 4:  # InstanceNorm2d, no affine, no running stats (building_blocks.py:26)
            return F.instance_norm(y, eps=eps)
        # InstanceNorm1d on the (B,T,C)-permuted tensor: per-(b,t) over C (building_blocks.py:50-51)
        return F.instance_norm(y.permute(0, 2, 1), eps=eps).permute(0, 2, 1)
    if norm == "BN":  # BatchNorm{1,2}d defaults: affine, track_running_stats (building_blocks.py:24,39)
        rm, rv = state[prefix + ".norm.running_mean"], state[prefix + ".norm.running_var"]
        out = F.batch_norm(y, rm, rv, state[prefix + ".norm.weight"], state[prefix + ".norm.bias"],
                           training, momentum, eps)
        if training:
            state[prefix + ".norm.num_batches_tracked"] += 1
        return out
    raise NotImplementedError(norm)


# bf16-STORAGE emulation (tests only).  The engine's bf16-storage mode (BASELINE config 4's arithmetic) keeps the
# audio encoder's activations, conv outputs and conv-operand weight copies as bf16 in HBM.  With BF16_EMULATION set, the float64 oracle rounds (to
# nearest even, as v_cvt_pk_bf16_f32 does) exactly where that path rounds -- and nowhere else -- so that what is left between the two is fp32
# summation order, not bf16 rounding:
#   block 0 (1 -> 64 channels, fused conv + norm + activation kernel): the block's OUTPUT;
#   blocks 1..7: the weights (the bf16 copy of the fp32 master), the conv output y as stored (the statistics come from the UNROUNDED accumulators),
#                the normalised + activated output (bf16; the last block writes fp32 for the fp32 1-D stage).
# Backward passes straight through the roundings (the gradient tensors' own bf16 storage is not emulated).
BF16_EMULATION = None


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _rb(x):
    return _RoundBF16.apply(x)


def conv_norm_act(x, state, prefix, stride, pad, norm, leaky, training, emu=None):
    w = state[prefix + ".conv.weight"]
    if emu in ("2d", "2d_last"):
        assert norm == "IN" and w.dim() == 4, "the bf16-storage emulation covers the InstanceNorm audio encoder"
        y = F.conv2d(x, _rb(w), None, stride, pad)
        mean, var = y.mean((2, 3), keepdim=True), y.var((2, 3), unbiased=False, keepdim=True)  # statistics of the unrounded outputs
        y = (_rb(y) - mean) / torch.sqrt(var + 1e-5)
        y = F.leaky_relu(y, 0.2) if leaky else F.relu(y)
        return y if emu == "2d_last" else _rb(y)
    y = F.conv2d(x, w, None, stride, pad) if w.dim() == 4 else F.conv1d(x, w, None, stride, pad)
    y = _norm(y, state, prefix, norm, training)
    y = F.leaky_relu(y, 0.2) if leaky else F.relu(y)
    return _rb(y) if emu == "l0" else y


def _block1d(x, state, prefix, down, norm, leaky, training):
    return conv_norm_act(x, state, prefix, 2 if down else 1, 1, norm, leaky, training)


# --------------------------------------------------------------------------------------
# Generator (generator.py)
# --------------------------------------------------------------------------------------
def audio_encoder(state, prefix, mel, num_frames, norm, leaky, training):
    """generator.py:39-43 -- mel (B,80,F) -> (B,256,num_frames)."""
    x = mel.unsqueeze(1)
    last = len(AUDIO_ENCODER_2D) - 1
    for i, (_, _, _, s, p) in enumerate(AUDIO_ENCODER_2D):
        emu = None if not BF16_EMULATION else ("l0" if i == 0 else ("2d_last" if i == last else "2d"))
        x = conv_norm_act(x, state, f"{prefix}.specgram_encoder_2d.{i // 2}.{i % 2}", s, p, norm, leaky, training, emu)
    x = F.interpolate(x, (1, num_frames), mode="bilinear")
    return x.squeeze(2)


def unet_1d(state, prefix, x, norm, leaky, training):
    """generator.py:70-85."""
    enc = []
    for name, down in UNET_ENC:
        x = _block1d(x, state, f"{prefix}.{name}", down, norm, leaky, training)
        enc.append(x)
    skips = enc[-2::-1]  # e5, e4, e3, e2, e1, e0
    for name, skip in zip(UNET_DEC, skips):
        x = _block1d(F.interpolate(x, skip.size(-1), mode="linear") + skip, state, f"{prefix}.{name}",
                     False, norm, leaky, training)
    return x


def generator(state, prefix, mel, num_frames, code, cfg, training):
    """SequenceGeneratorCNN.forward, generator.py:106-117 -> (B,num_frames,2,K)."""
    g = cfg.VOICE2POSE.GENERATOR
    x = audio_encoder(state, prefix + ".audio_encoder", mel, num_frames, g.NORM, g.LEAKY_RELU, training)
    if g.CLIP_CODE.DIMENSION is not None:
        x = torch.cat([x, code.unsqueeze(2).repeat(1, 1, x.shape[-1])], 1)
    x = unet_1d(state, prefix + ".unet", x, g.NORM, g.LEAKY_RELU, training)
    for i in range(4):
        x = _block1d(x, state, f"{prefix}.decoder.{i}", False, g.NORM, g.LEAKY_RELU, training)
    x = F.conv1d(x, state[prefix + ".decoder.4.weight"], state[prefix + ".decoder.4.bias"])
    return x.permute(0, 2, 1).reshape(-1, num_frames, 2, cfg.DATASET.NUM_LANDMARKS)


# --------------------------------------------------------------------------------------
# Pose VAE (autoencoder.py) and discriminator (discriminator.py)
# --------------------------------------------------------------------------------------
def pose_seq_encoder(state, prefix, poses, cfg, training):
    """autoencoder.py:27-35 -- (B,T,2,K) -> mu (B,D), logvar (B,D)."""
    a = cfg.POSE2POSE.AUTOENCODER
    x = poses.reshape(poses.shape[0], poses.shape[1], -1).permute(0, 2, 1)
    for i, down in enumerate(POSE_ENC_DOWN):
        x = _block1d(x, state, f"{prefix}.blocks.{i}", down, a.NORM, a.LEAKY_RELU, training)
    x = F.interpolate(x, 1).squeeze(-1)  # nearest -> first time step
    return x[:, 0::2], x[:, 1::2]


def pose_seq_decoder(state, prefix, code, cfg, training):
    """autoencoder.py:59-69 -- (B,D) -> (B,2K,64)."""
    a = cfg.POSE2POSE.AUTOENCODER
    x = F.interpolate(code.unsqueeze(-1), 2)
    for name in UNET_DEC:
        x = _block1d(F.interpolate(x, x.shape[-1] * 2, mode="linear"), state, f"{prefix}.{name}", False,
                     a.NORM, a.LEAKY_RELU, training)
    for i in range(4):
        x = _block1d(x, state, f"{prefix}.blocks.{i}", False, a.NORM, a.LEAKY_RELU, training)
    return F.conv1d(x, state[prefix + ".blocks.4.weight"], state[prefix + ".blocks.4.bias"])


def autoencoder(state, prefix, poses, num_frames, cfg, training, eps):
    """autoencoder.py:79-92 with the reparameterisation noise ``eps`` injected."""
    mu, logvar = pose_seq_encoder(state, prefix + ".encoder", poses, cfg, training)
    code = mu + torch.exp(0.5 * logvar) * eps
    x = pose_seq_decoder(state, prefix + ".decoder", code, cfg, training)
    x = x.permute(0, 2, 1).reshape(-1, num_frames, 2, cfg.DATASET.NUM_LANDMARKS)
    return x, mu, logvar


def discriminator(state, prefix, x, cfg, training):
    """discriminator.py:19-23 -- (B,T',2,K) -> patch scores (B,T'')."""
    leaky = cfg.VOICE2POSE.POSE_DISCRIMINATOR.LEAKY_RELU
    x = x.reshape(x.size(0), x.size(1), -1).transpose(1, 2)
    for i, (_, _, _, s, p) in enumerate(DISC_LAYERS):
        x = conv_norm_act(x, state, f"{prefix}.seq.{i}", s, p, "BN", leaky, training)
    x = F.conv1d(x, state[prefix + ".seq.3.weight"], state[prefix + ".seq.3.bias"], 1, 1)
    return x.squeeze(1)


# --------------------------------------------------------------------------------------
# Dataset-side transforms that run inside every train step (gesture_dataset.py)
# --------------------------------------------------------------------------------------
HEAD_ROOT, HAND_ROOT_L, HAND_ROOT_R = 39, 6, 3  # gesture_dataset.py:42-45


def parted_to_global(p):
    """gesture_dataset.py:147-155 (in place on xy of a (...,2,121) tensor)."""
    idx = list(range(9, HEAD_ROOT)) + list(range(HEAD_ROOT + 1, 79))
    p[..., :2, idx] = p[..., :2, idx] + p[..., :2, HEAD_ROOT, None]
    p[..., :2, 79:100] = p[..., :2, 79:100] + p[..., :2, HAND_ROOT_L, None]
    p[..., :2, 100:121] = p[..., :2, 100:121] + p[..., :2, HAND_ROOT_R, None]
    return p


def _stat_shape(t, kp):
    k = kp.shape[-1]
    if t.dim() == 1:
        return t.reshape(1, 2, k)
    if t.dim() == 2:
        return t.reshape(kp.shape[0], 1, 2, k)
    raise NotImplementedError


def denormalize(kp, stat):
    """gesture_dataset.py:193-211 (collated float64 stats promote the result to float64)."""
    return kp * _stat_shape(stat["std"], kp) + _stat_shape(stat["mean"], kp)


def normalize(kp, stat):
    """gesture_dataset.py:173-191."""
    return (kp - _stat_shape(stat["mean"], kp)) / _stat_shape(stat["std"], kp)


def get_final_results(poses, stat, hierarchical=True):
    """gesture_dataset.py:213-220."""
    poses = denormalize(poses, stat)
    if hierarchical:
        poses = parted_to_global(poses)
    scale = stat["scale_factor"].reshape(stat["scale_factor"].shape[0], 1, 1, -1)
    return poses * scale


def transform_normalized_parted2global(poses, stat_parted, stat_global):
    """gesture_dataset.py:222-236 (1-D float32 stats of the batch's first speaker)."""
    poses = denormalize(poses, stat_parted)
    poses = parted_to_global(poses)
    return normalize(poses, stat_global)


def evaluate_step(pred, gt):
    """voice2pose.py:412-430 -- L2 distance and normalised lip-sync error."""
    l2 = torch.norm(pred - gt, p=2, dim=2)
    lip_p = torch.norm(pred[:, :, :, 75] - pred[:, :, :, 71], p=2, dim=-1)
    lip_g = torch.norm(gt[:, :, :, 75] - gt[:, :, :, 71], p=2, dim=-1)
    den = lip_g.max(-1, keepdim=True).values + 1e-4
    return {"L2_dist": l2.mean(), "lip_sync_error_n": torch.abs(lip_p / den - lip_g / den).mean()}


# --------------------------------------------------------------------------------------
# Voice2PoseModel.forward (voice2pose.py:84-210), training-mode branch
# --------------------------------------------------------------------------------------
def clip_code_kl(code, lam):
    """voice2pose.py:147-157; returns None when any batch variance is exactly zero."""
    mu = code.mean(dim=0)
    var = code.var(dim=0)
    if not bool((var != 0).all()):
        return None
    return 0.5 * (-torch.log(var) + mu ** 2 + var - 1).mean() * lam


def voice2pose_forward(state, batch, cfg, training=True, stats_s2g=None):
    """Returns (losses, results).  ``state`` keys are Voice2PoseModel's state_dict names.
    BN running statistics in ``state`` are updated in place exactly as the module would."""
    g = cfg.VOICE2POSE.GENERATOR
    audio, poses_gt, idx = batch["audio"], batch["poses"], batch["clip_index"]
    num_frames = int(batch["num_frames"][0])
    code = state["clips_code"][idx] if g.CLIP_CODE.DIMENSION is not None else None
    mel = mel_spectrogram(audio, state.get("mel_transfm.spectrogram.window"), state.get("mel_transfm.mel_scale.fb"))
    pred = generator(state, "netG", mel, num_frames, code, cfg, training)
    results = {"poses_pred_batch": pred, "condition_code": code, "poses_gt_batch": poses_gt}
    losses = {}
    reg = (torch.abs(pred - poses_gt) * g.LAMBDA_REG).mean()  # voice2pose.py:141-142
    losses["G_reg_loss"] = reg
    g_loss = reg.clone()
    if code is not None:
        kl = clip_code_kl(code, g.LAMBDA_CLIP_KL)
        if kl is not None:
            losses["G_clipcode_kl_loss"] = kl
            g_loss = g_loss + kl
    losses["G_loss"] = g_loss
    if cfg.VOICE2POSE.POSE_ENCODER.NAME is not None:  # voice2pose.py:160-176; module stays in train mode
        with torch.no_grad():
            if cfg.DATASET.HIERARCHICAL_POSE:
                e_pred, e_gt = pred, poses_gt
            else:
                e_pred = transform_normalized_parted2global(pred, *stats_s2g)
                e_gt = transform_normalized_parted2global(poses_gt, *stats_s2g)
            mu_p, lv_p = pose_seq_encoder(state, "pose_encoder", e_pred, cfg, training)
            mu_g, lv_g = pose_seq_encoder(state, "pose_encoder", e_gt, cfg, training)
        results.update(mu_pred=mu_p, mu_gt=mu_g, logvar_pred=lv_p, logvar_gt=lv_g)
    d = cfg.VOICE2POSE.POSE_DISCRIMINATOR
    if d.NAME is not None:  # voice2pose.py:179-208
        real, fake = poses_gt, pred
        if d.MOTION:
            real = real[:, 1:] - real[:, :-1]
            fake = fake[:, 1:] - fake[:, :-1]
        s_real = discriminator(state, "netD_pose", real, cfg, training)
        s_fake = discriminator(state, "netD_pose", fake, cfg, training)
        s_fake_det = discriminator(state, "netD_pose", fake.detach(), cfg, training)
        g_gan = F.mse_loss(s_fake, torch.ones_like(s_fake)) * d.LAMBDA_GAN
        losses["G_pose_gan_loss"] = g_gan
        losses["G_loss"] = g_loss + g_gan
        d_loss = (F.mse_loss(s_real, torch.ones_like(s_real)) + F.mse_loss(s_fake_det, torch.zeros_like(s_fake_det))) * d.LAMBDA_GAN
        losses.update(D_pose_gan_loss=d_loss, pose_score_fake=s_fake.mean(), pose_score_real=s_real.mean())
    return losses, results


def pose2pose_forward(state, batch, cfg, eps, training=True):
    """Pose2PoseModel.forward, pose2pose.py:41-89 (mel is computed upstream and discarded; omitted)."""
    poses_gt = batch["poses"]
    num_frames = int(batch["num_frames"][0])
    pred, mu, logvar = autoencoder(state, "ae", poses_gt, num_frames, cfg, training, eps)
    reg = (torch.abs(pred - poses_gt) * cfg.POSE2POSE.LAMBDA_REG).mean()
    kl = 0.5 * (-logvar + mu ** 2 + torch.exp(logvar) - 1).mean() * cfg.POSE2POSE.LAMBDA_KL
    return ({"reg_loss": reg, "kl_loss": kl, "loss": reg + kl},
            {"poses_pred_batch": pred, "poses_gt_batch": poses_gt, "clip_code_mu": mu, "clip_code_logvar": logvar})


# --------------------------------------------------------------------------------------
# State construction (deterministic, numpy PCG64 -- regenerable on the GPU box)
# --------------------------------------------------------------------------------------
def _fill_conv_block(state, rng, prefix, cin, cout, ksize, norm, dtype):
    import numpy as np
    shape = (cout, cin) + tuple(ksize)
    fan_in = cin * int(np.prod(ksize))
    std = math.sqrt(2.0 / fan_in)  # kaiming_normal_, building_blocks.py:44
    state[prefix + ".conv.weight"] = torch.from_numpy(rng.standard_normal(shape) * std).to(dtype)
    if norm == "BN":
        # perturbed affine so that gamma/beta paths are exercised (module init would be 1/0)
        state[prefix + ".norm.weight"] = torch.from_numpy(1.0 + 0.1 * rng.standard_normal(cout)).to(dtype)
        state[prefix + ".norm.bias"] = torch.from_numpy(0.1 * rng.standard_normal(cout)).to(dtype)
        state[prefix + ".norm.running_mean"] = torch.zeros(cout, dtype=dtype)
        state[prefix + ".norm.running_var"] = torch.ones(cout, dtype=dtype)
        state[prefix + ".norm.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def _fill_head(state, rng, prefix, cin, cout, k, dtype):
    bound = 1.0 / math.sqrt(cin * k)  # nn.Conv1d default init bound
    state[prefix + ".weight"] = torch.from_numpy(rng.uniform(-bound, bound, (cout, cin, k))).to(dtype)
    state[prefix + ".bias"] = torch.from_numpy(rng.uniform(-bound, bound, (cout,))).to(dtype)


def fill_generator(state, rng, prefix, cfg, dtype=torch.float32):
    g = cfg.VOICE2POSE.GENERATOR
    for i, (ci, co, k, _, _) in enumerate(AUDIO_ENCODER_2D):
        _fill_conv_block(state, rng, f"{prefix}.audio_encoder.specgram_encoder_2d.{i // 2}.{i % 2}", ci, co, k, g.NORM, dtype)
    d = g.CLIP_CODE.DIMENSION or 0
    for name, down in UNET_ENC:
        _fill_conv_block(state, rng, f"{prefix}.unet.{name}", 256 + (d if name == "e0" else 0), 256,
                         (4,) if down else (3,), g.NORM, dtype)
    for name in UNET_DEC:
        _fill_conv_block(state, rng, f"{prefix}.unet.{name}", 256, 256, (3,), g.NORM, dtype)
    for i in range(4):
        _fill_conv_block(state, rng, f"{prefix}.decoder.{i}", 256, 256, (3,), g.NORM, dtype)
    _fill_head(state, rng, f"{prefix}.decoder.4", 256, cfg.DATASET.NUM_LANDMARKS * 2, 1, dtype)


def fill_pose_encoder(state, rng, prefix, cfg, dtype=torch.float32):
    a = cfg.POSE2POSE.AUTOENCODER
    cin = cfg.DATASET.NUM_LANDMARKS * 2
    for i, down in enumerate(POSE_ENC_DOWN):
        co = a.CODE_DIM * 2 if i == len(POSE_ENC_DOWN) - 1 else 256
        _fill_conv_block(state, rng, f"{prefix}.blocks.{i}", cin if i == 0 else 256, co, (4,) if down else (3,), a.NORM, dtype)


def fill_pose_decoder(state, rng, prefix, cfg, dtype=torch.float32):
    a = cfg.POSE2POSE.AUTOENCODER
    for j, name in enumerate(UNET_DEC):
        _fill_conv_block(state, rng, f"{prefix}.{name}", a.CODE_DIM if j == 0 else 256, 256, (3,), a.NORM, dtype)
    for i in range(4):
        _fill_conv_block(state, rng, f"{prefix}.blocks.{i}", 256, 256, (3,), a.NORM, dtype)
    _fill_head(state, rng, f"{prefix}.blocks.4", 256, cfg.DATASET.NUM_LANDMARKS * 2, 1, dtype)


def fill_discriminator(state, rng, prefix, cfg, dtype=torch.float32):
    cin0 = cfg.DATASET.NUM_LANDMARKS * 2
    for i, (ci, co, k, _, _) in enumerate(DISC_LAYERS):
        _fill_conv_block(state, rng, f"{prefix}.seq.{i}", cin0 if ci is None else ci, co, (k,), "BN", dtype)
    _fill_head(state, rng, f"{prefix}.seq.3", 1024, 1, 3, dtype)


def make_voice2pose_state(cfg, n_clips, seed=0, dtype=torch.float32, code_std=0.0):
    """Full Voice2PoseModel state (voice2pose.py:22-82 member order).  ``code_std`` > 0 fills the
    clip-code table with N(0, code_std) instead of the module's zeros (exercises the KL branch)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    st = {}
    g = cfg.VOICE2POSE.GENERATOR
    if g.CLIP_CODE.DIMENSION is not None:
        tab = rng.standard_normal((n_clips, g.CLIP_CODE.DIMENSION)) * code_std
        st["clips_code"] = torch.from_numpy(tab).to(dtype)
    st["mel_transfm.spectrogram.window"] = mel_window(dtype)
    st["mel_transfm.mel_scale.fb"] = mel_filterbank(dtype)
    fill_generator(st, rng, "netG", cfg, dtype)
    if cfg.VOICE2POSE.POSE_ENCODER.NAME is not None:
        fill_pose_encoder(st, rng, "pose_encoder", cfg, dtype)
    if cfg.VOICE2POSE.POSE_DISCRIMINATOR.NAME is not None:
        fill_discriminator(st, rng, "netD_pose", cfg, dtype)
    return st


def make_pose2pose_state(cfg, n_clips, seed=0, dtype=torch.float32):
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    st = {"clip_code_mu": torch.zeros(n_clips, cfg.POSE2POSE.AUTOENCODER.CODE_DIM, dtype=dtype),
          "clip_code_logvar": torch.zeros(n_clips, cfg.POSE2POSE.AUTOENCODER.CODE_DIM, dtype=dtype)}
    fill_pose_encoder(st, rng, "ae.encoder", cfg, dtype)
    fill_pose_decoder(st, rng, "ae.decoder", cfg, dtype)
    return st


def make_batch(batch_size, n_clips, step=0, seed=1, dtype=torch.float32, num_kp=121, num_frames=64, audio_len=68266):
    """Seeded synthetic batch with the GestureDataset.__getitem__ field layout (gesture_dataset.py:107-119)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64([seed, step]))
    audio = (0.1 * rng.standard_normal((batch_size, audio_len))).astype(np.float32)
    poses = rng.standard_normal((batch_size, num_frames, 2, num_kp)).astype(np.float32)
    idx = (np.arange(batch_size) + step * batch_size) % n_clips
    mean = rng.standard_normal((batch_size, 2 * num_kp)) * 20.0
    std = rng.uniform(2.0, 30.0, (batch_size, 2 * num_kp))
    scale = rng.uniform(0.8, 1.3, (batch_size,))
    return {
        "audio": torch.from_numpy(audio).to(dtype),
        "poses": torch.from_numpy(poses).to(dtype),
        "clip_index": torch.from_numpy(idx.astype(np.int64)),
        "num_frames": torch.full((batch_size,), num_frames, dtype=torch.int64),
        "speaker": ["synthetic"] * batch_size,
        "speaker_stat": {"scale_factor": torch.from_numpy(scale), "mean": torch.from_numpy(mean),
                         "std": torch.from_numpy(std)},
    }


# --------------------------------------------------------------------------------------
# Train step (voice2pose.py:281-312) on an oracle-held state
# --------------------------------------------------------------------------------------
_NON_PARAM_SUFFIX = ("running_mean", "running_var", "num_batches_tracked", "spectrogram.window", "mel_scale.fb")


class OracleVoice2Pose:
    """Holds parameters as autograd leaves plus torch.optim.Adam instances laid out like
    Voice2Pose.setup_optimizer (voice2pose.py:244-279); ``train_step`` replays :281-312."""

    def __init__(self, cfg, state, lr=None):
        self.cfg = cfg
        self.state = state
        lr = cfg.TRAIN.LR if lr is None else lr
        g = cfg.VOICE2POSE.GENERATOR
        for k, v in state.items():
            if v.is_floating_point() and not k.endswith(_NON_PARAM_SUFFIX):
                if k == "clips_code" and (g.CLIP_CODE.EXTERNAL_CODE or not g.CLIP_CODE.TRAIN):
                    continue
                v.requires_grad_(True)
        self.opt = {"optimizerG": torch.optim.Adam([v for k, v in state.items() if k.startswith("netG.") and v.requires_grad],
                                                   lr=lr, weight_decay=cfg.TRAIN.WD)}
        if cfg.VOICE2POSE.POSE_DISCRIMINATOR.NAME is not None:
            self.opt["optimizerD_pose"] = torch.optim.Adam(
                [v for k, v in state.items() if k.startswith("netD_pose.") and v.requires_grad], lr=lr)
        if g.CLIP_CODE.DIMENSION is not None and not g.CLIP_CODE.EXTERNAL_CODE:
            self.opt["optimizerClipCode"] = torch.optim.Adam([state["clips_code"]], lr=lr * g.CLIP_CODE.LR_SCALING)

    def train_step(self, batch, stats_s2g=None):
        losses, results = voice2pose_forward(self.state, batch, self.cfg, True, stats_s2g)
        return self._finish_step(losses, results, batch)

    def _finish_step(self, losses, results, batch):
        hier = self.cfg.DATASET.HIERARCHICAL_POSE
        fin_p = get_final_results(results["poses_pred_batch"].detach(), batch["speaker_stat"], hier)
        fin_g = get_final_results(results["poses_gt_batch"].detach(), batch["speaker_stat"], hier)
        losses.update(evaluate_step(fin_p, fin_g))
        if "optimizerClipCode" in self.opt:
            self.opt["optimizerClipCode"].zero_grad()
        self.opt["optimizerG"].zero_grad()
        losses["G_loss"].backward(retain_graph=True)
        if "optimizerClipCode" in self.opt:
            self.opt["optimizerClipCode"].step()
        self.opt["optimizerG"].step()
        if "optimizerD_pose" in self.opt:
            self.opt["optimizerD_pose"].zero_grad()
            losses["D_pose_gan_loss"].backward()
            self.opt["optimizerD_pose"].step()
        results["final_pred"], results["final_gt"] = fin_p, fin_g
        return {k: v.detach() for k, v in losses.items()}, results


class OraclePose2Pose:
    """Pose2Pose.train_step (pose2pose.py:124-149) on an oracle-held state; ``eps`` is the reparameterisation noise."""

    def __init__(self, cfg, state, lr=None):
        self.cfg, self.state = cfg, state
        for k, v in state.items():
            if k.startswith("ae.") and v.is_floating_point() and not k.endswith(_NON_PARAM_SUFFIX):
                v.requires_grad_(True)
        self.opt = torch.optim.Adam([v for k, v in state.items() if k.startswith("ae.") and v.requires_grad],
                                    lr=cfg.TRAIN.LR if lr is None else lr, weight_decay=cfg.TRAIN.WD)

    def train_step(self, batch, eps):
        losses, results = pose2pose_forward(self.state, batch, self.cfg, eps, True)
        fin_p = get_final_results(results["poses_pred_batch"].detach(), batch["speaker_stat"], self.cfg.DATASET.HIERARCHICAL_POSE)
        fin_g = get_final_results(results["poses_gt_batch"].detach(), batch["speaker_stat"], self.cfg.DATASET.HIERARCHICAL_POSE)
        losses.update(evaluate_step(fin_p, fin_g))
        idx = batch["clip_index"]
        self.state["clip_code_mu"][idx] = results["clip_code_mu"].detach()  # pose2pose.py:135-137
        self.state["clip_code_logvar"][idx] = results["clip_code_logvar"].detach()
        self.opt.zero_grad()
        losses["loss"].backward()
        self.opt.step()
        return {k: v.detach() for k, v in losses.items()}, results
