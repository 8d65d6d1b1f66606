"""Generate golden vectors by running the REFERENCE's own modules (imported from /root/reference).

Run in the authoring container only:  ``python tests/golden/make_golden.py``.
Nothing of the reference is copied: its modules are imported under a small shim (stubs for the
optional packages this image lacks, identity ``.cuda()``), driven with deterministic numpy-PCG64
inputs / weights produced by ``oracle.sdt_oracle`` builders, and only inputs' seeds + OUTPUTS are
written to ``tests/golden/*.npz``.  The torchaudio MelSpectrogram stand-in used for the reference
model run is the oracle's restatement (torchaudio 0.7.0 is not available) -> mel parity unpinned.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import sdt_oracle as O  # noqa: E402


def install_shim():
    for name in ["cv2", "ffmpeg", "librosa", "yacs", "yacs.config"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb

    ta = types.ModuleType("torchaudio")
    tr = types.ModuleType("torchaudio.transforms")

    class _Spec(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("window", O.mel_window())

    class _Scale(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("fb", O.mel_filterbank())

    class MelSpectrogram(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            assert kw == dict(win_length=400, hop_length=160, n_fft=512, f_min=55, f_max=7500.0, n_mels=80), kw
            self.spectrogram = _Spec()
            self.mel_scale = _Scale()

        def forward(self, x):
            return O.mel_spectrogram(x, self.spectrogram.window, self.mel_scale.fb)

    tr.MelSpectrogram = MelSpectrogram
    ta.transforms = tr
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tr
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    np.float = float
    sys.path.insert(0, REF)


def ref_dataset(hierarchical):
    from core.datasets.gesture_dataset import GestureDataset
    ds = object.__new__(GestureDataset)
    ds.cfg = types.SimpleNamespace(NUM_LANDMARKS=121, HIERARCHICAL_POSE=hierarchical)
    ds.root_node, ds.hand_root_l, ds.hand_root_r, ds.head_root = 1, 6, 3, 39
    return ds


def clone_state(st):
    return {k: v.clone() for k, v in st.items()}


def sl(t, n=64):
    """Deterministic strided sample of a tensor (first n of a stride walk) + sum + abs-sum."""
    f = t.detach().reshape(-1).double()
    step = max(1, f.numel() // n)
    return np.concatenate([f[::step][:n].numpy(), [f.sum().item(), f.abs().sum().item()]])


def run_reference_model(cfg_name, B, n_clips, steps, code_std, dtype=torch.float32):
    """Replays voice2pose.py:281-309 around the imported Voice2PoseModel."""
    from core.pipelines.voice2pose import Voice2PoseModel
    import core.datasets.speakers_stat as SS
    cfg = O.cfg_named(cfg_name)
    st0 = O.make_voice2pose_state(cfg, n_clips, seed=0, dtype=dtype, code_std=code_std)
    ext = None
    if cfg.VOICE2POSE.GENERATOR.CLIP_CODE.EXTERNAL_CODE:  # sdt_vae: fixed codes come from a pose-VAE checkpoint file
        import tempfile
        ext = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal((n_clips, 32)).astype(np.float32))
        f = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
        torch.save({"model_state_dict": {"module.clip_code_mu": ext, "module.clip_code_logvar": torch.zeros_like(ext)}}, f.name)
        cfg.VOICE2POSE.POSE_ENCODER.AE_CHECKPOINT = f.name
        st0.pop("clips_code")
    model = Voice2PoseModel(cfg, None, n_clips)
    if dtype == torch.float64:
        model = model.double()
    missing = model.load_state_dict(clone_state(st0), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.train()
    ds = ref_dataset(cfg.DATASET.HIERARCHICAL_POSE)
    optG = torch.optim.Adam(model.netG.parameters(), lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    optC = torch.optim.Adam([model.clips_code], lr=cfg.TRAIN.LR) if cfg_name == "voice2pose_sdt_bp" else None
    optD = torch.optim.Adam(model.netD_pose.parameters(), lr=cfg.TRAIN.LR) if hasattr(model, "netD_pose") else None
    out = {}
    for step in range(steps):
        batch = O.make_batch(B, n_clips, step=step, seed=1, dtype=dtype)
        if not cfg.DATASET.HIERARCHICAL_POSE:
            batch["speaker"] = ["oliver"] * B
        losses, results = model(batch, ds)
        fin_p = ds.get_final_results(results["poses_pred_batch"].detach(), batch["speaker_stat"])
        fin_g = ds.get_final_results(results["poses_gt_batch"].detach(), batch["speaker_stat"])
        L2 = torch.norm(fin_p - fin_g, p=2, dim=2)
        # evaluate_step is a Trainer method (needs a CUDA device to construct); replay its body via the oracle
        # only for the two metric scalars -- the transforms above are the reference's own.
        metrics = O.evaluate_step(fin_p, fin_g)
        assert torch.allclose(metrics["L2_dist"], L2.mean())
        if optC is not None:
            optC.zero_grad()
        optG.zero_grad()
        losses["G_loss"].backward(retain_graph=True)
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        if optC is not None:
            optC.step()
        optG.step()
        if optD is not None:
            optD.zero_grad()
            losses["D_pose_gan_loss"].backward()
            for k, p in model.named_parameters():
                if k.startswith("netD_pose."):
                    grads["Dstep:" + k] = p.grad.detach().clone()
            optD.step()
        for k, v in losses.items():
            out[f"s{step}/loss/{k}"] = np.array(v.item())
        for k, v in metrics.items():
            out[f"s{step}/metric/{k}"] = np.array(v.item())
        out[f"s{step}/pred"] = (results["poses_pred_batch"].detach().numpy().astype(np.float64 if dtype == torch.float64 else np.float32)
                                if step == 0 else sl(results["poses_pred_batch"]))
        out[f"s{step}/final_pred"] = sl(fin_p)
        for k in ("mu_pred", "mu_gt", "logvar_pred", "logvar_gt"):
            out[f"s{step}/{k}"] = results[k].detach().numpy()
        for k, g in grads.items():  # every step (step 0 was the round-1 fixture; later steps feed the fp64-calibrated checks)
            out[f"s{step}/grad/{k}"] = sl(g)
    if ext is not None:
        os.unlink(cfg.VOICE2POSE.POSE_ENCODER.AE_CHECKPOINT)
        assert torch.equal(model.clips_code, ext)
    for k, v in model.state_dict().items():
        out[f"final/{k}"] = sl(v) if v.is_floating_point() else np.array(v.item())
    if "clips_code" in model.state_dict():
        out["final_full/clips_code_rows"] = model.state_dict()["clips_code"][: 3 * B].numpy()
    return out


def run_reference_pose2pose(B, n_clips, steps, dtype=torch.float32):
    """Replays pose2pose.py:124-149 around the imported Pose2PoseModel (reparameterisation noise injected)."""
    from core.pipelines.pose2pose import Pose2PoseModel
    cfg = O.cfg_named("pose2pose")
    st0 = O.make_pose2pose_state(cfg, n_clips, seed=0, dtype=dtype)
    model = Pose2PoseModel(cfg, None, n_clips)
    if dtype == torch.float64:
        model = model.double()
    st0["mel_transfm.spectrogram.window"] = O.mel_window(dtype)
    st0["mel_transfm.mel_scale.fb"] = O.mel_filterbank(dtype)
    r = model.load_state_dict(clone_state(st0), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    model.train()
    ds = ref_dataset(True)
    opt = torch.optim.Adam(model.ae.parameters(), lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    out = {}
    real_randn = torch.randn
    for step in range(steps):
        batch = O.make_batch(B, n_clips, step=step, seed=1, dtype=dtype)
        eps = torch.from_numpy(np.random.Generator(np.random.PCG64([2, step])).standard_normal((B, 32)).astype(np.float32)).to(dtype)
        torch.randn = lambda *a, **k: eps.clone()
        try:
            losses, results = model(batch)
        finally:
            torch.randn = real_randn
        fin_p = ds.get_final_results(results["poses_pred_batch"].detach(), batch["speaker_stat"])
        fin_g = ds.get_final_results(results["poses_gt_batch"].detach(), batch["speaker_stat"])
        metrics = O.evaluate_step(fin_p, fin_g)
        idx = batch["clip_index"]
        model.clip_code_mu[idx] = results["clip_code_mu"].detach()
        model.clip_code_logvar[idx] = results["clip_code_logvar"].detach()
        opt.zero_grad()
        losses["loss"].backward(retain_graph=True)
        for k, p in model.named_parameters():
            out[f"s{step}/grad/{k}"] = sl(p.grad)
        opt.step()
        for k, v in losses.items():
            out[f"s{step}/loss/{k}"] = np.array(v.item())
        for k, v in metrics.items():
            out[f"s{step}/metric/{k}"] = np.array(v.item())
        out[f"s{step}/pred"] = sl(results["poses_pred_batch"])
        if step == 0:  # the full step-0 prediction: lets a test count L1 sign decisions that differ between two fp32 runs
            out["s0/pred_full"] = results["poses_pred_batch"].detach().numpy()
        out[f"s{step}/mu"] = results["clip_code_mu"].detach().numpy()
    for k, v in model.state_dict().items():
        out[f"final/{k}"] = sl(v) if v.is_floating_point() else np.array(v.item())
    return out


def main():
    install_shim()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from core.networks import get_model

    # ---- registry error convention (core/networks/__init__.py:14-19)
    try:
        get_model("nope")
        raise SystemExit("expected KeyError")
    except KeyError:
        pass

    # ---- (1) module-level forwards, B=2 --------------------------------------------------------
    mods = {}
    B = 2
    batch = O.make_batch(B, 16, step=0, seed=1)
    mel = O.mel_spectrogram(batch["audio"])
    mods["mel"] = mel[:1].numpy()  # restatement output (unpinned); kept to detect drift
    rng = np.random.Generator(np.random.PCG64(7))
    code = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32))
    for norm in ("IN", "BN"):
        for dim in (None, 32):
            cfg = O.default_cfg(**{"VOICE2POSE.GENERATOR.NORM": norm, "VOICE2POSE.GENERATOR.CLIP_CODE.DIMENSION": dim})
            st = {}
            O.fill_generator(st, np.random.Generator(np.random.PCG64(3)), "netG", cfg)
            net = get_model("SequenceGeneratorCNN")(cfg)
            net.load_state_dict({k[len("netG."):]: v.clone() for k, v in st.items()}, strict=True)
            net.train()
            out = net(mel, 64, code if dim else None)
            tag = f"G_{norm}_{dim}"
            mods[tag] = out.detach().numpy()
            if norm == "BN":
                mods[tag + "/rm0"] = net.state_dict()["audio_encoder.specgram_encoder_2d.0.0.norm.running_mean"].numpy()
                mods[tag + "/rv_last"] = net.state_dict()["decoder.3.norm.running_var"].numpy()
            if norm == "IN" and dim == 32:  # variable-length inference (demo path), T != 64
                net.eval()
                mods[tag + "/T40"] = net(mel[:, :, :300], 40, code).detach().numpy()
    cfg = O.cfg_named("voice2pose_s2g")
    st = {}
    O.fill_discriminator(st, np.random.Generator(np.random.PCG64(4)), "netD_pose", cfg)
    net = get_model("PoseSequenceDiscriminator")(cfg)
    net.load_state_dict({k[len("netD_pose."):]: v.clone() for k, v in st.items()}, strict=True)
    net.train()
    motion = batch["poses"][:, 1:] - batch["poses"][:, :-1]
    mods["D_leaky"] = net(motion).detach().numpy()
    cfg_relu = O.default_cfg()
    net2 = get_model("PoseSequenceDiscriminator")(cfg_relu)
    net2.load_state_dict({k[len("netD_pose."):]: v.clone() for k, v in st.items()}, strict=True)
    net2.train()
    mods["D_relu"] = net2(motion).detach().numpy()

    cfg = O.cfg_named("pose2pose")
    st = {}
    O.fill_pose_encoder(st, np.random.Generator(np.random.PCG64(5)), "enc", cfg)
    net = get_model("PoseSeqEncoder")(cfg)
    net.load_state_dict({k[len("enc."):]: v.clone() for k, v in st.items()}, strict=True)
    net.train()
    mu, lv = net(batch["poses"])
    mods["PoseEnc/mu"], mods["PoseEnc/logvar"] = mu.detach().numpy(), lv.detach().numpy()
    net.eval()
    mu, lv = net(batch["poses"])
    mods["PoseEnc_eval/mu"] = mu.detach().numpy()

    st = O.make_pose2pose_state(cfg, 16, seed=6)
    net = get_model("Autoencoder")(cfg)
    net.load_state_dict({k[len("ae."):]: v.clone() for k, v in st.items() if k.startswith("ae.")}, strict=True)
    net.train()
    eps = torch.from_numpy(np.random.Generator(np.random.PCG64(2)).standard_normal((B, 32)).astype(np.float32))
    real_randn = torch.randn
    torch.randn = lambda *a, **k: eps.clone()
    try:
        out, mu, lv = net(batch["poses"], 64)
    finally:
        torch.randn = real_randn
    mods["AE/out"], mods["AE/mu"], mods["AE/logvar"] = out.detach().numpy(), mu.detach().numpy(), lv.detach().numpy()

    # ---- (2) dataset-side transforms in float64 (gesture_dataset.py:193-236) -------------------
    ds = ref_dataset(True)
    fin = ds.get_final_results(batch["poses"].clone(), batch["speaker_stat"])
    assert fin.dtype == torch.float64
    mods["final_results"] = fin.numpy()
    import core.datasets.speakers_stat as SS
    sp = {"parted": SS.SPEAKERS_STAT_121_parted["oliver"], "global": SS.SPEAKERS_STAT_121["oliver"]}
    np.savez_compressed(os.path.join(HERE, "speaker_stat_oliver.npz"),
                        parted_mean=sp["parted"]["mean"], parted_std=sp["parted"]["std"],
                        parted_scale=np.array(sp["parted"]["scale_factor"]),
                        global_mean=sp["global"]["mean"], global_std=sp["global"]["std"],
                        global_scale=np.array(sp["global"]["scale_factor"]))
    mods["p2g_oliver"] = ds.transform_normalized_parted2global(batch["poses"].clone(), ["oliver"] * B).numpy()
    np.savez_compressed(os.path.join(HERE, "modules_B2.npz"), **mods)

    # ---- (3) full model: losses, grads, 3-step Adam trajectories -------------------------------
    traj = {}
    for name, Bm, code_std in (("voice2pose_sdt_bp", 4, 0.5), ("voice2pose_sdt_bp_zero", 4, 0.0),
                               ("voice2pose_s2g", 4, 0.0), ("voice2pose_sdt_vae", 4, 0.0)):
        cfg_name = name.replace("_zero", "")
        res = run_reference_model(cfg_name, Bm, 16, 3, code_std)
        for k, v in res.items():
            traj[f"{name}/{k}"] = v
        print(name, {k: float(v) for k, v in res.items() if "/loss/" in k})
    res = run_reference_pose2pose(4, 16, 3)
    for k, v in res.items():
        traj[f"pose2pose/{k}"] = v
    print("pose2pose", {k: float(v) for k, v in res.items() if "/loss/" in k})
    # fp64 reference run of the same sdt_bp trajectory: sets the tolerance floor
    res = run_reference_model("voice2pose_sdt_bp", 4, 16, 1, 0.5, dtype=torch.float64)
    for k, v in res.items():
        if k.startswith("s0/"):
            traj[f"voice2pose_sdt_bp_f64/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "trajectories_B4.npz"), **traj)

    # ---- (4) float64 runs of the REFERENCE for every config, all 3 steps: the "truth" that calibrates every fp32 tolerance
    # (|HIP - f64| is compared with |reference fp32 - f64| per loss, per prediction and per gradient tensor, per step)
    t64 = {}
    for name, Bm, code_std in (("voice2pose_sdt_bp", 4, 0.5), ("voice2pose_sdt_bp_zero", 4, 0.0),
                               ("voice2pose_s2g", 4, 0.0), ("voice2pose_sdt_vae", 4, 0.0)):
        res = run_reference_model(name.replace("_zero", ""), Bm, 16, 3, code_std, dtype=torch.float64)
        for k, v in res.items():
            if k.startswith("final"):
                continue
            t64[f"{name}/{k}"] = v.astype(np.float64) if k.endswith("/pred") else v
        print(name, "f64", {k: float(v) for k, v in res.items() if "/loss/G_loss" in k})
    res = run_reference_pose2pose(4, 16, 3, dtype=torch.float64)
    for k, v in res.items():
        if not k.startswith("final"):
            t64[f"pose2pose/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "trajectories_B4_f64.npz"), **t64)

    # ---- (5) Frechet gesture distance (core/utils/fgd.py:59-64) on seeded code sets -------------------------------
    from core.utils.fgd import compute_fgd
    fg = {}
    rng = np.random.Generator(np.random.PCG64(11))
    for tag, n, d in (("n200_d32", 200, 32), ("n64_d64", 64, 64), ("n500_d32", 500, 32)):
        a = rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, d)
        b = rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, d) + 0.3 * rng.standard_normal(d)
        mix = rng.standard_normal((d, d)) / np.sqrt(d)
        b = b @ (np.eye(d) + 0.3 * mix)  # correlated dimensions: a non-diagonal covariance product
        fg[tag + "/a"], fg[tag + "/b"] = a, b
        fg[tag + "/fgd_ab"] = compute_fgd(a, b).numpy()  # float32, as the reference returns it
        fg[tag + "/fgd_aa"] = compute_fgd(a, a).numpy()
    a = rng.standard_normal((20, 32))  # fewer samples than dimensions: singular covariances (the jitter / complex branch)
    b = rng.standard_normal((20, 32)) + 0.5
    fg["n20_d32/a"], fg["n20_d32/b"], fg["n20_d32/fgd_ab"] = a, b, compute_fgd(a, b).numpy()
    np.savez_compressed(os.path.join(HERE, "fgd.npz"), **fg)
    for f in ("modules_B2.npz", "trajectories_B4.npz", "trajectories_B4_f64.npz", "fgd.npz", "speaker_stat_oliver.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
