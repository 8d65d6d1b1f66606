"""GPU parity of the reference-shaped modules and of Voice2Pose.train_step against
(a) the fixtures produced by the REFERENCE's own modules (tests/golden/*.npz) and (b) the CPU oracle.
fp32 tolerances are stated per check; gradient tolerances are calibrated on the reference's own
fp32-vs-fp64 discrepancy stored in the fixtures."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import sdt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def sl(t, n=64):
    f = t.detach().reshape(-1).double().cpu()
    step = max(1, f.numel() // n)
    return np.concatenate([f[::step][:n].numpy(), [f.sum().item(), f.abs().sum().item()]])


def relmax(got, ref):
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    ref = ref.detach().double().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all()
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def check(name, got, ref, tol):
    """`tol`: the stated fp32 tolerance of this quantity; the check is ALSO held to 10x the error recorded in tests/golden/margins.json
    (conftest.calibrated_bound), so that a 10x numerical regression fails even where the stated tolerance is generous."""
    from conftest import calibrated_bound
    e = relmax(got, ref)
    bound = calibrated_bound(name, e, tol)
    print("  %-46s rel-max-err %.3e (tol %.1e, held to %.1e)" % (name, e, tol, bound))
    assert e < bound, "%s: %.3e >= %.1e (stated tolerance %.1e)" % (name, e, bound, tol)


@pytest.fixture(scope="module")
def batch2():
    return O.make_batch(2, 16, step=0, seed=1)


def _to_dev(state, prefix):
    return {k[len(prefix):]: v.clone() for k, v in state.items() if k.startswith(prefix)}


@pytest.mark.parametrize("norm", ["IN", "BN"])
@pytest.mark.parametrize("dim", [None, 32])
def test_generator_vs_reference_fixture(batch2, golden_modules, norm, dim):
    from speechdrivestemplates_amd.core.networks import get_model
    cfg = O.default_cfg(**{"VOICE2POSE.GENERATOR.NORM": norm, "VOICE2POSE.GENERATOR.CLIP_CODE.DIMENSION": dim})
    st = {}
    O.fill_generator(st, np.random.Generator(np.random.PCG64(3)), "netG", cfg)
    net = get_model("SequenceGeneratorCNN")(cfg)
    net.load_state_dict(_to_dev(st, "netG."), strict=True)
    net.to(DEV).train()
    mel = O.mel_spectrogram(batch2["audio"]).to(DEV)
    code = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).standard_normal((2, 32)).astype(np.float32)).to(DEV)
    out = net(mel, 64, code if dim else None)
    tag = "G_%s_%s" % (norm, dim)
    assert out.shape == (2, 64, 2, 121)
    check(tag + " vs reference", out, golden_modules[tag], 2e-4)
    if norm == "BN":
        sd = net.state_dict()
        check(tag + " running_mean L0", sd["audio_encoder.specgram_encoder_2d.0.0.norm.running_mean"], golden_modules[tag + "/rm0"], 1e-4)
        check(tag + " running_var decoder.3", sd["decoder.3.norm.running_var"], golden_modules[tag + "/rv_last"], 1e-4)
        assert int(sd["unet.e3.norm.num_batches_tracked"]) == 1
    if norm == "IN" and dim == 32:  # variable-length inference (demo path)
        net.eval()
        with torch.no_grad():
            out = net(mel[:, :, :300].contiguous(), 40, code)
        check(tag + " T=40 vs reference", out, golden_modules[tag + "/T40"], 2e-4)
        # sub-module API with the reference's logical shapes
        with torch.no_grad():
            enc = net.audio_encoder(mel, 64)
            assert enc.shape == (2, 256, 64)
            un = net.unet(torch.cat([enc, code.unsqueeze(2).repeat(1, 1, 64)], 1))
            assert un.shape == (2, 256, 64)


def test_discriminator_pose_encoder_autoencoder_vs_reference(batch2, golden_modules):
    from speechdrivestemplates_amd.core.networks import get_model
    poses = batch2["poses"].to(DEV)
    motion = poses[:, 1:] - poses[:, :-1]
    st = {}
    O.fill_discriminator(st, np.random.Generator(np.random.PCG64(4)), "netD_pose", O.cfg_named("voice2pose_s2g"))
    for cfg, tag in ((O.cfg_named("voice2pose_s2g"), "D_leaky"), (O.default_cfg(), "D_relu")):
        net = get_model("PoseSequenceDiscriminator")(cfg)
        net.load_state_dict(_to_dev(st, "netD_pose."), strict=True)
        net.to(DEV).train()
        check(tag + " vs reference", net(motion), golden_modules[tag], 2e-4)
    cfg = O.cfg_named("pose2pose")
    st = {}
    O.fill_pose_encoder(st, np.random.Generator(np.random.PCG64(5)), "enc", cfg)
    net = get_model("PoseSeqEncoder")(cfg)
    net.load_state_dict(_to_dev(st, "enc."), strict=True)
    net.to(DEV).train()
    mu, lv = net(poses)
    check("PoseSeqEncoder mu (train)", mu, golden_modules["PoseEnc/mu"], 5e-4)
    check("PoseSeqEncoder logvar (train)", lv, golden_modules["PoseEnc/logvar"], 5e-4)
    net.eval()
    with torch.no_grad():
        mu, _ = net(poses)
    check("PoseSeqEncoder mu (eval, running stats)", mu, golden_modules["PoseEnc_eval/mu"], 5e-4)
    st = O.make_pose2pose_state(cfg, 16, seed=6)
    net = get_model("Autoencoder")(cfg)
    net.load_state_dict(_to_dev(st, "ae."), strict=True)
    net.to(DEV).train()
    eps = torch.from_numpy(np.random.Generator(np.random.PCG64(2)).standard_normal((2, 32)).astype(np.float32)).to(DEV)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: eps.clone()
    try:
        out, mu, lv = net(poses, 64)
    finally:
        torch.randn = real_randn
    check("Autoencoder out", out, golden_modules["AE/out"], 5e-4)
    check("Autoencoder mu", mu, golden_modules["AE/mu"], 5e-4)


def _make_pipeline(cfg_name, n_clips, code_std, extra_opts=()):
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import gesture_dataset as gd
    from speechdrivestemplates_amd.core.pipelines import get_pipeline
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", cfg_name + ".yaml"))
    from speechdrivestemplates_amd import ops as _ops
    # the pipeline owns its storage mode (cfg.SYS.STORAGE -> Trainer.knobs, applied at every step): a test that selected a mode with
    # ops.set_storage() before building its pipeline gets a pipeline configured for that mode
    cfg.merge_from_list(["DATASET.NAME", "SyntheticGestureDataset", "DATASET.SYNTHETIC_CLIPS", n_clips, "SYS.LOG_INTERVAL", 10 ** 9,
                         "SYS.STORAGE", _ops.STORAGE, "SYS.CHAIN1D", bool(_ops.CHAIN1D), "SYS.CONV_F32_SPLIT", bool(_ops.F32_SPLIT)] + list(extra_opts))
    cfg.freeze()
    sp = np.load(os.path.join(GOLDEN, "speaker_stat_oliver.npz"))
    gd.register_speaker_stat("oliver",
                             parted={"mean": sp["parted_mean"], "std": sp["parted_std"], "scale_factor": float(sp["parted_scale"])},
                             global_={"mean": sp["global_mean"], "std": sp["global_std"], "scale_factor": float(sp["global_scale"])})
    pipe = get_pipeline(cfg.PIPELINE_TYPE)(cfg)
    pipe.num_train_samples = n_clips
    pipe.train_dataset = gd.SyntheticGestureDataset(cfg=cfg, num_clips=n_clips)
    ocfg = O.cfg_named(cfg_name)
    if cfg_name == "pose2pose":
        st = O.make_pose2pose_state(ocfg, n_clips, seed=0)
        st["mel_transfm.spectrogram.window"], st["mel_transfm.mel_scale.fb"] = O.mel_window(), O.mel_filterbank()
        pipe.setup_model(cfg, state_dict={"module." + k: v for k, v in st.items()})
    else:
        st = O.make_voice2pose_state(ocfg, n_clips, seed=0, code_std=code_std)
        kw = {}
        if ocfg.VOICE2POSE.GENERATOR.CLIP_CODE.EXTERNAL_CODE:  # sdt_vae: fixed codes, same seed as make_golden.py
            st.pop("clips_code")
            kw["external_codes"] = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal((n_clips, 32)).astype(np.float32))
        pipe.setup_model(cfg, state_dict={"module." + k: v for k, v in st.items()}, **kw)
    pipe.setup_optimizer()
    pipe.model.train()
    return pipe, cfg


@pytest.mark.parametrize("name,code_std", [("voice2pose_sdt_bp", 0.5), ("voice2pose_sdt_bp_zero", 0.0), ("voice2pose_s2g", 0.0),
                                           ("voice2pose_sdt_vae", 0.0)])
def test_train_step_trajectory_vs_reference(golden_traj, name, code_std):
    cfg_name = name.replace("_zero", "")
    pipe, cfg = _make_pipeline(cfg_name, 16, code_std)
    g = {k[len(name) + 1:]: v for k, v in golden_traj.items() if k.startswith(name + "/")}
    g64 = {k[len("voice2pose_sdt_bp_f64/"):]: v for k, v in golden_traj.items() if k.startswith("voice2pose_sdt_bp_f64/")}
    for step in range(3):
        batch = O.make_batch(4, 16, step=step, seed=1)
        if cfg_name == "voice2pose_s2g":
            batch["speaker"] = ["oliver"] * 4
        losses, results = pipe.forward_backward(batch, want_final=True)
        if step == 0:
            check(name + " step0 pred vs reference", results["poses_pred_normalized"], g["s0/pred"], 2e-4)
            grads = {k: p.grad.detach().clone() for k, p in pipe.model.named_parameters() if p.grad is not None}
            worst = 0.0
            for k, ref in g.items():
                if not k.startswith("s0/grad/") or k.startswith("s0/grad/Dstep:"):
                    continue
                pk = k[len("s0/grad/"):]
                got = sl(grads[pk])[:64]
                # reference fp32 vs reference fp64 on the same 64 samples = the noise floor of this gradient
                floor = np.abs(ref[:64] - g64["s0/grad/" + pk][:64]).max() if (name == "voice2pose_sdt_bp" and "s0/grad/" + pk in g64) else 0.0
                scale = max(np.abs(ref[:64]).max(), 1e-12)
                err = np.abs(got - ref[:64]).max()
                # without a stored fp64 floor (the _zero / s2g runs) allow the reference's own documented fp32
                # noise on early-layer weight gradients (up to 1e-2 of max|grad|)
                tol = max(4.0 * floor, 2e-2 * scale)
                worst = max(worst, err / scale)
                assert err <= tol, "grad %s: err %.3e > tol %.3e (scale %.3e, ref fp32-vs-fp64 floor %.3e)" % (pk, err, tol, scale, floor)
            print("  %-46s worst grad-slice rel err %.3e" % (name + " step0 grads vs reference", worst))
        pipe.optimizer_updates(losses)
        torch.cuda.synchronize()
        for k in [x for x in g if x.startswith("s%d/loss/" % step)]:
            lk = k.split("/")[-1]
            # after Adam updates: 3e-3 (sign-like first steps turn fp32 summation-order noise into lr-sized weight differences);
            # the adversarial terms sit behind TWO updated networks (G and D) and measured 1e-3..3e-3 run to run -> 1e-2
            tol = 2e-4 if step == 0 else (1e-2 if ("gan" in lk or "score" in lk) else 3e-3)
            check("%s s%d %s" % (name, step, lk), losses[lk], g[k], tol)
        if name == "voice2pose_sdt_bp_zero":
            assert float(losses["G_clipcode_kl_loss"]) == 0.0 and int(results["kl_valid"]) == 0  # device-side skip
        check("%s s%d L2_dist" % (name, step), losses["L2_dist"], g["s%d/metric/L2_dist" % step], 1e-4)
        check("%s s%d lip_sync" % (name, step), losses["lip_sync_error_n"], g["s%d/metric/lip_sync_error_n" % step], 1e-3)
        check("%s s%d final_pred" % (name, step), sl(results["poses_pred_batch"]), g["s%d/final_pred" % step], 5e-4)
        for k in ("mu_pred", "mu_gt", "logvar_pred", "logvar_gt"):
            # after the first Adam update (|dw| = lr regardless of |g|: sign noise of near-zero gradients) the two
            # fp32 runs are different-but-equivalent trajectories; features of the prediction drift accordingly
            check("%s s%d %s" % (name, step, k), results[k], g["s%d/%s" % (step, k)], 5e-3 if step == 0 else 1e-1)
    # state after 3 Adam steps: weights (each element moved by <= lr per step; sign-noise on near-zero grads bounds
    # the achievable agreement at ~2*lr*steps), BN buffers and counters
    lr, steps = 1e-4, 3
    sd = pipe.model.state_dict()
    for k, v in sd.items():
        ref = g["final/" + k]
        if not v.is_floating_point():
            assert int(v) == int(ref), k
            continue
        got = sl(v)
        # BatchNorm running statistics integrate the (slightly drifting) activations of steps 1-2: 1e-2 relative
        rel = 1e-2 if ("running_" in k) else 2e-3
        assert np.abs(got[:-2] - ref[:-2]).max() <= 2.2 * lr * steps + rel * np.abs(ref[:-2]).max(), k
        assert abs(got[-1] - ref[-1]) <= rel * ref[-1] + 2.2 * lr * steps * v.numel(), k  # abs-sum of the whole tensor
    if "clips_code" in sd:
        # dense Adam moves a touched element by ~(1, 1.67, 2.2)*lr*sign(g): an element whose gradient is at the
        # fp32 noise level may legitimately flip sign between two fp32 implementations, so require agreement on
        # >= 97 % of the elements and bound the rest by the largest possible excursion
        got = sd["clips_code"][:12].double().cpu().numpy()
        ref = g["final_full/clips_code_rows"].astype(np.float64)
        diff = np.abs(got - ref)
        agree = diff <= 2e-2 * np.abs(ref).max()
        print("  %-46s %.1f %% of elements agree, worst %.3e" % (name + " clips_code (dense Adam)", 100 * agree.mean(), diff.max()))
        assert agree.mean() >= 0.97 and diff.max() <= 2 * 2.3 * lr * steps
        untouched = sd["clips_code"][12:]
        ref_untouched = O.make_voice2pose_state(O.cfg_named(cfg_name), 16, seed=0, code_std=code_std)["clips_code"][12:]
        assert torch.equal(untouched.cpu(), ref_untouched), "rows with zero gradient and zero moments must not move"


def test_pose2pose_trajectory_vs_reference(golden_traj):
    """Config 5: Pose2Pose.train_step (pose VAE, BN everywhere, analytic KL) against the reference-generated fixture."""
    pipe, cfg = _make_pipeline("pose2pose", 16, 0.0)
    g = {k[len("pose2pose/"):]: v for k, v in golden_traj.items() if k.startswith("pose2pose/")}
    real_randn = torch.randn
    for step in range(3):
        batch = O.make_batch(4, 16, step=step, seed=1)
        eps = torch.from_numpy(np.random.Generator(np.random.PCG64([2, step])).standard_normal((4, 32)).astype(np.float32)).to(DEV)
        torch.randn = lambda *a, **k: eps.clone()
        try:
            losses, results = pipe.forward_backward(batch)
        finally:
            torch.randn = real_randn
        if step == 0:
            grads = {k: p.grad.detach().clone() for k, p in pipe.model.named_parameters() if p.grad is not None}
            worst = 0.0
            for k, ref in g.items():
                if k.startswith("s0/grad/"):
                    got = sl(grads[k[len("s0/grad/"):]])[:-2]
                    scale = max(np.abs(ref[:-2]).max(), 1e-12)
                    err = np.abs(got - ref[:-2]).max()
                    worst = max(worst, err / scale)
                    # 16 training-mode BatchNorm layers in sequence amplify fp32 summation-order noise in the weight
                    # gradients (the reference's own fp32-vs-fp64 gap is of this order)
                    assert err <= 5e-2 * scale, "grad %s: %.3e (scale %.3e)" % (k, err, scale)
            print("  %-46s worst grad-slice rel err %.3e" % ("pose2pose step0 grads vs reference", worst))
        pipe.optimizer_updates(losses)
        for k in ("reg_loss", "kl_loss", "loss"):
            check("pose2pose s%d %s" % (step, k), losses[k], g["s%d/loss/%s" % (step, k)], 2e-4 if step == 0 else 3e-3)
        check("pose2pose s%d L2_dist" % step, losses["L2_dist"], g["s%d/metric/L2_dist" % step], 1e-3)
        check("pose2pose s%d pred" % step, sl(results["poses_pred_batch"]), g["s%d/pred" % step], 2e-4 if step == 0 else 2e-2)
        check("pose2pose s%d mu" % step, results["clip_code_mu"], g["s%d/mu" % step], 5e-4 if step == 0 else 5e-2)
    sd = pipe.model.state_dict()
    for k, v in sd.items():
        ref = g["final/" + k]
        if not v.is_floating_point():
            assert int(v) == int(ref), k
            continue
        got = sl(v)
        rel = 1e-2 if ("running_" in k or "clip_code" in k) else 2e-3
        assert np.abs(got[:-2] - ref[:-2]).max() <= 2.2e-4 * 3 + rel * max(np.abs(ref[:-2]).max(), 1e-3), k


def test_checkpoint_roundtrip_and_flat_buffers():
    pipe, cfg = _make_pipeline("voice2pose_sdt_bp", 16, 0.5)
    optg = pipe.optimizers["optimizerG"]
    # every generator parameter is a view into the flat buffer, in the kernels' (Cout,taps,Cin) layout
    w = pipe.model.netG.unet.e2.conv.weight
    assert w.data_ptr() >= optg.flat_param.data_ptr() and w.data_ptr() < optg.flat_param.data_ptr() + optg.flat_param.numel() * 4
    assert w.permute(0, 2, 1).is_contiguous() and w.grad.permute(0, 2, 1).is_contiguous()
    batch = O.make_batch(4, 16, step=0, seed=1)
    losses, _ = pipe.forward_backward(batch)
    pipe.optimizer_updates(losses)
    ckpt = pipe.checkpoint_dict(1, 1)
    assert set(ckpt) == {"epoch", "step", "model_state_dict", "optimizerG_state_dict", "optimizerClipCode_state_dict"}
    assert all(k.startswith("module.") for k in ckpt["model_state_dict"])
    assert ckpt["optimizerG_state_dict"]["state"][0]["exp_avg"].shape == pipe.optimizers["optimizerG"].params[0].shape
    pipe2, _ = _make_pipeline("voice2pose_sdt_bp", 16, 0.0)
    pipe2.setup_model(cfg, state_dict=ckpt["model_state_dict"])
    pipe2.optimizers, pipe2.schedulers = {}, {}
    pipe2.setup_optimizer(checkpoint=ckpt)
    pipe2.model.train()
    b1 = O.make_batch(4, 16, step=1, seed=1)
    l1, _ = pipe.forward_backward(b1)
    pipe.optimizer_updates(l1)
    l2, _ = pipe2.forward_backward(b1)
    pipe2.optimizer_updates(l2)
    check("resumed run reproduces G_loss", l2["G_loss"], l1["G_loss"], 1e-5)
    check("resumed run reproduces weights", pipe2.model.netG.decoder[4].weight, pipe.model.netG.decoder[4].weight, 1e-5)


def test_aux_stream_overlap_is_bit_identical():
    """ops.OVERLAP_AUX runs the no-grad pose-encoder passes on a side stream: results, BN buffers and weights after
    two steps must equal the single-stream run exactly (same kernels, same order per stream, deterministic split-K)."""
    from speechdrivestemplates_amd import ops
    outs = []
    prev = (ops.OVERLAP_AUX, ops.OVERLAP_DW)
    for flag in (False, True):
        ops.OVERLAP_AUX, ops.OVERLAP_DW = flag, False  # isolate the aux overlap from the weight-gradient side stream
        try:
            pipe, _ = _make_pipeline("voice2pose_sdt_vae", 16, 0.0)  # no atomically-accumulated code-table gradient
            for step in range(2):
                losses, results = pipe.forward_backward(O.make_batch(4, 16, step=step, seed=1))
                pipe.optimizer_updates(losses)
            torch.cuda.synchronize()
            outs.append((results["mu_pred"].clone(), results["logvar_gt"].clone(), losses["G_reg_loss"].clone(),
                         pipe.model.pose_encoder.blocks[3].norm.running_var.clone(), results["poses_pred_normalized"].clone()))
        finally:
            ops.OVERLAP_AUX, ops.OVERLAP_DW = prev
    for a, b in zip(*outs):
        # (weight gradients are ordered slab sums since round 3; the two schedules still differ in which BatchNorm running statistics the
        # pose-encoder passes see first, hence a small tolerance on everything downstream)
        check("overlap vs single stream", b, a, 5e-5)


def test_deferred_weight_gradients_cover_every_layer():
    """ops.DEFER_SMALL_DW: the 1-D stage's weight-gradient launches are recorded during backward and enqueued as one batch
    on the side stream from the hook at the audio encoder's output.  Nothing may be left behind when forward_backward
    returns, the hook must be what flushes them, and the gradients must equal the inline launches (the same ordered slab sums)."""
    from speechdrivestemplates_amd import ops
    grads, seen = [], []
    prev, orig_flush = ops.DEFER_SMALL_DW, ops.flush_deferred_dw

    def spy():
        seen.append(len(ops._DEFERRED))
        orig_flush()

    for flag in (False, True):
        ops.DEFER_SMALL_DW = flag
        ops.flush_deferred_dw = spy
        try:
            pipe, _ = _make_pipeline("voice2pose_sdt_bp", 16, 0.5)
            del seen[:]
            losses, _ = pipe.forward_backward(O.make_batch(4, 16, step=0, seed=1))
            assert not ops._DEFERRED and not ops._DEFER_ON[0]
            if flag:
                # first flush = the hook at the encoder output: U-Net (12) + decoder (5) convolutions, none from the 2-D stage
                # (at this batch size some Conv2d layers fall under the side stream's size threshold too: later flushes)
                assert seen and seen[0] == 17, seen
            else:
                assert all(n == 0 for n in seen), seen
            torch.cuda.synchronize()
            grads.append(pipe.optimizers["optimizerG"].flat_grad.clone())
        finally:
            ops.DEFER_SMALL_DW, ops.flush_deferred_dw = prev, orig_flush
    check("deferred vs inline weight gradients", grads[1], grads[0], 2e-5)


@pytest.mark.parametrize("B", [3, 33])
def test_ragged_batches_match_the_oracle(B):
    """Edge batches through the WHOLE train step (the plans are tuned for 32 clips; the fixtures hold 2 and 4): 3 clips = ragged tiles in every launch,
    33 = one clip more than a full batch.  Step-0 losses and float64 metrics against the CPU oracle, prediction to the forward tolerance; a batch
    of ONE clip raises where the reference's unbiased variance over one code turns the loss into NaN (voice2pose.py:152-155)."""
    from speechdrivestemplates_amd import ops
    n_clips = 40
    pipe, cfg = _make_pipeline("voice2pose_sdt_bp", n_clips, 0.5)
    ocfg = O.cfg_named("voice2pose_sdt_bp")
    state = O.make_voice2pose_state(ocfg, n_clips, seed=0, code_std=0.5)
    eng = O.OracleVoice2Pose(ocfg, state)
    batch = O.make_batch(B, n_clips, step=0, seed=5)
    losses, results = pipe.forward_backward(batch)
    pipe.optimizer_updates(losses)
    ref_losses, ref_results = eng.train_step(batch)
    torch.cuda.synchronize()
    for k in ("G_reg_loss", "G_clipcode_kl_loss", "G_loss", "L2_dist", "lip_sync_error_n"):
        a, b = float(losses[k].detach()), float(ref_losses[k])
        assert abs(a - b) <= 2e-6 * abs(b) + 1e-7, (B, k, a, b)
    check("ragged batch B=%d prediction vs oracle" % B, results["poses_pred_normalized"], ref_results["poses_pred_batch"], 2e-4)
    assert ops.streamk_error_codes() == {}
    if B == 3:
        with pytest.raises(RuntimeError, match="at least 2 clips"):
            pipe.forward_backward(O.make_batch(1, n_clips, step=1, seed=5))


@pytest.mark.parametrize("name,code_std,tol", [("voice2pose_sdt_vae", 0.0, 2e-5), ("voice2pose_sdt_bp", 0.5, 1e-3)])
def test_overlapped_gradient_exchange_single_rank(name, code_std, tol):
    """The bucketed, backward-overlapped all-reduce path (dp.GradReducer + the post-encoder hook) exercised over RCCL with
    a 1-rank process group (SDT_DP_FORCE): the exchange must cover every gradient element exactly once and leave the
    training trajectory unchanged."""
    import torch.distributed as dist
    from speechdrivestemplates_amd import dp
    hist = {}
    for forced in (False, True):
        if forced:
            os.environ["SDT_DP_FORCE"] = "1"
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
        try:
            pipe, _ = _make_pipeline(name, 16, code_std)
            assert pipe.reducer.active == forced
            if forced:
                assert pipe.model.netG.post_encoder_grad_hook is not None
                calls = []
                orig = pipe.reducer.launch

                def spy(opt, lo=0, hi=None, _orig=orig, _calls=calls, _optg=pipe.optimizers["optimizerG"]):
                    if opt is _optg:
                        _calls.append((lo, opt.flat_grad.numel() if hi is None else hi))
                    return _orig(opt, lo, hi)

                pipe.reducer.launch = spy
            out = []
            for step in range(2):
                losses, results = pipe.forward_backward(O.make_batch(4, 16, step=step, seed=1))
                pipe.optimizer_updates(losses)
                torch.cuda.synchronize()
                out.append(float(losses["G_loss"]))
            hist[forced] = (out, pipe.model.netG.unet.e0.conv.weight.detach().clone())
            if forced:
                n = pipe.optimizers["optimizerG"].flat_grad.numel()
                per_step = calls[:len(calls) // 2]
                # buckets in backward order: [U-Net + decoder], [L5..L7], [L3..L4], [L1..L2], [L0]; together the whole buffer once
                assert len(per_step) == 5 and per_step[0][1] == n and per_step[-1][0] == 0, per_step
                for a, b in zip(per_step[:-1], per_step[1:]):
                    assert b[1] == a[0] and b[0] < a[0], per_step
        finally:
            if forced:
                os.environ.pop("SDT_DP_FORCE", None)
                dist.destroy_process_group()
                from speechdrivestemplates_amd import ops as _ops
                assert _ops.SK_RESERVED_SLOTS == dp.RESERVED_SLOTS  # an active reducer makes the backward stream-K launches leave room for the collective
                pipe.close()                                        # ... a process-wide setting that goes away with the last active reducer
                assert _ops.SK_RESERVED_SLOTS == 0 and dp.active_reducers() == 0
    for a, b in zip(hist[False][0], hist[True][0]):
        assert abs(a - b) <= tol * abs(a), hist
    assert torch.isfinite(hist[True][1]).all()


@pytest.mark.parametrize("mode,tol_pred,tol_loss", [("bf16", 4e-2, 2e-2), ("bf16x6", 2e-4, 3e-3)])
def test_conv_math_modes_train_steps(golden_traj, mode, tol_pred, tol_loss):
    """BASELINE config 4 asks for bf16: the opt-in 'bf16' conv arithmetic (operands rounded to bf16, fp32 accumulate,
    everything else fp32) must track the reference trajectory within bf16 tolerances; 'bf16x6' (fp32-equivalent
    products on the bf16 MFMA) within the fp32 tolerances of the default path."""
    from speechdrivestemplates_amd import ops
    ops.set_conv_math(mode)
    try:
        pipe, _ = _make_pipeline("voice2pose_sdt_bp", 16, 0.5)
        g = golden_traj
        for step in range(3):
            losses, results = pipe.forward_backward(O.make_batch(4, 16, step=step, seed=1))
            pipe.optimizer_updates(losses)
            ref_loss = float(g["voice2pose_sdt_bp/s%d/loss/G_loss" % step])
            got_loss = float(losses["G_loss"].detach())
            assert abs(got_loss - ref_loss) <= tol_loss * abs(ref_loss), (mode, step, got_loss, ref_loss)
            if step == 0:
                ref = torch.from_numpy(g["voice2pose_sdt_bp/s0/pred"])
                got = results["poses_pred_normalized"].detach().cpu()
                err = ((got - ref).abs().max() / ref.abs().max()).item()
                assert err <= tol_pred, (mode, err)
    finally:
        ops.set_conv_math("f32")


def test_full_size_step_properties_b32():
    """BASELINE config 2/3 size (32 clips per GPU), where the CPU oracle takes minutes: size-independent properties of the
    per-sample-normalised sdt generator and its L1 loss instead --
      * permuting the clips of a batch permutes the predictions (no cross-sample arithmetic in the generator), up to fp32
        summation order;
      * the gradient of a 32-clip batch is the mean of the gradients of its two 16-clip halves (what data parallelism
        relies on), up to fp32 summation noise."""
    def sub(batch, idx):
        out = {}
        for k, v in batch.items():
            if torch.is_tensor(v):
                out[k] = v[idx]
            elif isinstance(v, dict):
                out[k] = {kk: vv[idx] for kk, vv in v.items()}
            elif isinstance(v, list):
                out[k] = [v[i] for i in idx.tolist()]
            else:
                out[k] = v
        return out

    pipe, _ = _make_pipeline("voice2pose_sdt_vae", 64, 0.0)
    full = O.make_batch(32, 64, step=0, seed=5)
    optg = pipe.optimizers["optimizerG"]

    def grads_of(batch):
        losses, results = pipe.forward_backward(batch)
        torch.cuda.synchronize()
        return optg.flat_grad.detach().clone(), results["poses_pred_normalized"].detach().clone(), float(losses["G_reg_loss"].detach())

    g_full, pred_full, l_full = grads_of(full)
    perm = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).permutation(32))
    _, pred_perm, l_perm = grads_of(sub(full, perm))
    # not bit-for-bit: the InstanceNorm statistics come from the conv epilogue, whose fp32 partial sums depend on how a clip's
    # rows fall onto the 64-row tiles (its position in the batch); measured 2e-6
    check("B=32 permutation equivariance", pred_perm, pred_full[perm.to(pred_full.device)], 1e-5)
    assert abs(l_perm - l_full) <= 1e-6 * abs(l_full)
    g_a, _, l_a = grads_of(sub(full, torch.arange(0, 16)))
    g_b, _, l_b = grads_of(sub(full, torch.arange(16, 32)))
    assert abs(0.5 * (l_a + l_b) - l_full) <= 2e-6 * abs(l_full)
    half = 0.5 * (g_a + g_b)
    err = (half - g_full).abs().max().item() / g_full.abs().max().item()
    print("  B=32 gradient additivity: max|mean(halves) - full| / max|full| = %.3e" % err)
    assert err <= 2e-3, err  # early-layer weight gradients carry ~1e-3 relative fp32 summation noise


@pytest.mark.parametrize("cfg_name,code_std", [("voice2pose_sdt_bp", 0.5), ("voice2pose_s2g", 0.0)])
def test_hipgraph_replay_matches_eager(cfg_name, code_std):
    """graph.GraphedStep captures forward+backward+Adam -- the persistent stream-K launches included: their flags are lowered by their
    consumers, so a launch replays with the epoch it was captured with -- into one hipGraph; replayed steps must follow the eager run.
    s2g: the discriminator's second backward and optimiser step are inside the capture, and the parted -> global re-normalisation in front
    of the pose encoder runs on cached device copies of the speaker statistics and index tables (it used to copy them per call, which a
    capture refuses); that config is host-bound when enqueued launch by launch (11.6 ms of host work for 10.0 ms of GPU work per step)."""
    from speechdrivestemplates_amd.graph import GraphedStep
    runs = []
    for use_graph in (False, True):
        pipe, _ = _make_pipeline(cfg_name, 16, code_std)
        dev = pipe.model._device()
        gs = GraphedStep(pipe, warmup=1)
        hist = []
        for step in range(4):
            b = O.make_batch(4, 16, step=step, seed=1)
            b = {k: (v.to(dev) if torch.is_tensor(v) and k != "num_frames" else v) for k, v in b.items()}
            b["speaker_stat"] = {k: v.to(dev) for k, v in b["speaker_stat"].items()}
            if use_graph:
                losses = gs.run(b)
            else:
                losses, _ = pipe.forward_backward(b)
                pipe.optimizer_updates(losses)
            torch.cuda.synchronize()
            hist.append((float(losses["G_loss"]), float(losses["L2_dist"])))
        runs.append((hist, pipe.model.netG.decoder[4].weight.detach().clone(), int(pipe.optimizers["optimizerG"].state_dev[0])))
    (h0, w0, s0), (h1, w1, s1) = runs
    assert s0 == s1 == 4
    for i, ((a, b), (c, d)) in enumerate(zip(h0, h1)):
        # step 0 sees identical weights (atomics order only); later steps sit behind Adam updates that turn last-bit
        # gradient differences into lr-sized weight differences
        tol = 2e-5 if i == 0 else 2e-4
        assert abs(a - c) <= tol * abs(a) and abs(b - d) <= tol * abs(b), (h0, h1)
    # weight-gradient atomics make the two runs differ in the last bits; Adam's sign-like early steps amplify that to at
    # most 2*lr per step and element
    assert (w1 - w0).abs().max().item() <= 2 * 1e-4 * 4


def test_hipgraph_replay_of_the_pose2pose_step():
    """BASELINE config 5 (pose-sequence VAE): ~150 launches of a few microseconds; enqueued one by one the step is bound by the host (3.2 ms),
    replayed from a hipGraph it takes 1.34 ms (SYS.HIP_GRAPH; 9980 -> 23840 clips/s).  Replayed steps must follow the eager run: the random
    draw of the reparameterisation is part of the capture (torch's philox offsets are graph-safe), the clip-code buffers are updated in place."""
    from speechdrivestemplates_amd.graph import GraphedStep
    runs = []
    for use_graph in (False, True):
        torch.manual_seed(3)
        pipe, _ = _make_pipeline("pose2pose", 16, 0.0)
        dev = pipe.model.clip_code_mu.device
        gs = GraphedStep(pipe, warmup=1)
        hist = []
        for step in range(4):
            b = O.make_batch(4, 16, step=step, seed=1)
            b = {k: (v.to(dev) if torch.is_tensor(v) and k != "num_frames" else v) for k, v in b.items()}
            b["speaker_stat"] = {k: v.to(dev) for k, v in b["speaker_stat"].items()}
            if use_graph:
                losses = gs.run(b)
            else:
                losses, _ = pipe.forward_backward(b)
                pipe.optimizer_updates(losses)
            torch.cuda.synchronize()
            hist.append((float(losses["loss"]), float(losses["L2_dist"])))
        runs.append((hist, int(pipe.optimizers["optimizer"].state_dev[0])))
    (h0, s0), (h1, s1) = runs
    assert s0 == s1 == 4
    assert all(torch.isfinite(torch.tensor(h)).all() for h in (h0, h1))
    # the reparameterisation noise differs between an eager and a captured generator state, so the comparison is statistical: same scale, both falling
    assert abs(h0[0][1] - h1[0][1]) <= 0.2 * abs(h0[0][1]) and h1[-1][0] <= 1.05 * h1[0][0], (h0, h1)


def test_demo_step_variable_length():
    """Row f-4: 24 s of audio -> 360 frames through the same kernels (T != 64), code picked by DEMO.CODE_INDEX with
    interpolation towards CODE_INDEX_B (voice2pose.py:107-117,386-410), checked against the oracle in eval mode."""
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import gesture_dataset as gd
    from speechdrivestemplates_amd.core.pipelines import get_pipeline
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "voice2pose_sdt_bp.yaml"))
    cfg.merge_from_list(["DATASET.NAME", "SyntheticGestureDataset", "DEMO.CODE_INDEX", 3, "DEMO.CODE_INDEX_B", 5, "TEST.SAVE_NPZ", False])
    cfg.freeze()
    pipe = get_pipeline("Voice2Pose")(cfg)
    pipe.num_train_samples = 16
    pipe.test_dataset = gd.SyntheticGestureDataset(cfg=cfg, num_clips=16, split="val")
    ocfg = O.cfg_named("voice2pose_sdt_bp")
    st = O.make_voice2pose_state(ocfg, 16, seed=0, code_std=0.5)
    pipe.setup_model(cfg, state_dict={"module." + k: v for k, v in st.items()})
    L = int(360 * 16000 / 15)
    rng = np.random.Generator(np.random.PCG64(21))
    audio = torch.from_numpy((0.1 * rng.standard_normal((1, L))).astype(np.float32))
    stat = {"scale_factor": torch.tensor([1.1]), "mean": torch.from_numpy(rng.standard_normal((1, 242)) * 20.0),
            "std": torch.from_numpy(rng.uniform(2.0, 30.0, (1, 242)))}
    batch = {"audio": audio, "speaker": ["synthetic"], "clip_index": torch.tensor([0]), "num_frames": torch.tensor([360]),
             "speaker_stat": stat}
    res = pipe.demo_step(batch, interpolation_coeff=0.25)
    assert res["poses_pred_batch"].shape == (1, 360, 2, 121) and res["poses_pred_batch"].dtype == torch.float64
    code = st["clips_code"][3:4] * 0.75 + st["clips_code"][5:6] * 0.25
    pred = O.generator(st, "netG", O.mel_spectrogram(audio), 360, code, ocfg, False)
    ref = O.get_final_results(pred, stat, True)
    check("demo T=360 final poses vs oracle", res["poses_pred_batch"], ref, 2e-4)


def test_validate_loop_and_fgd():
    """Row f-1: eval-mode test_step over a small synthetic validation set + epoch-level FGD (voice2pose.py:333-384,432-446,
    trainer.py:407-427, core/utils/fgd.py)."""
    from speechdrivestemplates_amd.core.datasets import gesture_dataset as gd
    from speechdrivestemplates_amd.fgd import compute_fgd
    pipe, cfg = _make_pipeline("voice2pose_sdt_bp", 16, 0.5)
    pipe.test_dataset = gd.SyntheticGestureDataset(cfg=cfg, num_clips=8, split='val')
    pipe.test_dataloader = torch.utils.data.DataLoader(pipe.test_dataset, batch_size=4, shuffle=False)
    pipe.num_test_samples, pipe.num_test_batches = 8, 2
    # one training step first so that BN running statistics of the pose encoder are not the init values
    losses, _ = pipe.forward_backward(O.make_batch(4, 16, step=0, seed=1))
    pipe.optimizer_updates(losses)
    # record what every test_step saw (the batch and the randomly drawn code rows, voice2pose.py:119-120) so that the ORACLE can redo the
    # whole loop: per batch the eval-mode forward + float64 metrics, then the reference's aggregation (trainer.py:407-427: per-key sums of
    # loss x TEST.BATCH_SIZE over the batches / number of test samples; FGD over the concatenated pose-encoder features)
    seen = []
    orig_forward = pipe.model.forward

    def spy(batch, dataset, *a, **k):
        out_ = orig_forward(batch, dataset, *a, **k)
        seen.append((batch, out_[1]["condition_code"].detach().cpu().clone()))
        return out_
    pipe.model.forward = spy
    try:
        out = pipe.validate(pipe.test_dataloader, 1)
    finally:
        pipe.model.forward = orig_forward
    for k in ("G_reg_loss", "G_loss", "L2_dist", "lip_sync_error_n", "FGD_mu", "FGD_mu_logvar"):
        assert k in out and np.isfinite(float(out[k])), (k, out.get(k))
    assert not pipe.model.training  # validate() leaves the model in eval mode like the reference (trainer.py:409)
    assert len(seen) == 2
    ocfg = O.cfg_named("voice2pose_sdt_bp")
    st0 = {k: v.detach().cpu().clone() for k, v in pipe.model.state_dict().items()}
    sums, feats = {}, {k: [] for k in ("mu_pred", "mu_gt", "logvar_pred", "logvar_gt")}
    for batch, code in seen:
        B = code.shape[0]
        ob = {"audio": batch["audio"].cpu(), "poses": batch["poses"].cpu(), "clip_index": torch.arange(B),
              "num_frames": batch["num_frames"], "speaker_stat": {k: v.cpu() for k, v in batch["speaker_stat"].items()}}
        st = dict(st0, clips_code=code)  # row b of the table = the code the step drew for clip b
        with torch.no_grad():
            ol, ores = O.voice2pose_forward(st, ob, ocfg, training=False)
        fin_p = O.get_final_results(ores["poses_pred_batch"].double(), ob["speaker_stat"], True)
        fin_g = O.get_final_results(ob["poses"].double(), ob["speaker_stat"], True)
        ol.update(O.evaluate_step(fin_p, fin_g))
        for k, v in ol.items():
            sums[k] = sums.get(k, 0.0) + float(v) * cfg.TEST.BATCH_SIZE
        for k in feats:
            feats[k].append(ores[k].numpy())
    want = {k: v / pipe.num_test_samples for k, v in sums.items()}
    feats = {k: np.concatenate(v, 0) for k, v in feats.items()}
    want["FGD_mu"] = compute_fgd(feats["mu_pred"], feats["mu_gt"])
    want["FGD_mu_logvar"] = compute_fgd(np.concatenate([feats["mu_pred"], feats["logvar_pred"]], 1), np.concatenate([feats["mu_gt"], feats["logvar_gt"]], 1))
    for k in ("G_reg_loss", "G_clipcode_kl_loss", "G_loss", "L2_dist", "lip_sync_error_n"):
        check("validate() aggregate %s vs oracle loop" % k, torch.as_tensor(float(out[k])).reshape(1), torch.as_tensor(want[k]).reshape(1), 2e-4)
    for k in ("FGD_mu", "FGD_mu_logvar"):  # Frechet distance of 8 samples of 32 / 64 features: differences of covariance square roots
        assert abs(float(out[k]) - want[k]) <= 2e-3 * max(abs(want[k]), 1.0), (k, float(out[k]), want[k])
    # eval forward against the oracle in eval mode (running-stat BN, random code rows replaced by fixed ones)
    pipe.model.eval()
    batch = O.make_batch(4, 16, step=5, seed=1)
    st = {k: v.detach().cpu().clone() for k, v in pipe.model.state_dict().items()}
    torch.manual_seed(3)
    with torch.no_grad():
        _, res = pipe.model(batch, pipe.test_dataset)
    code = res["condition_code"].cpu()
    mel = O.mel_spectrogram(batch["audio"])
    ref = O.generator(st, "netG", mel, 64, code, O.cfg_named("voice2pose_sdt_bp"), False)
    check("eval-mode prediction vs oracle", res["poses_pred_batch"], ref, 2e-4)
    mu_ref, _ = O.pose_seq_encoder(st, "pose_encoder", batch["poses"], O.cfg_named("voice2pose_sdt_bp"), False)
    check("eval-mode pose-encoder features vs oracle", res["mu_gt"], mu_ref, 5e-4)
    a, b = np.random.default_rng(0).standard_normal((200, 32)), np.random.default_rng(1).standard_normal((200, 32)) + 0.1
    assert abs(compute_fgd(a, a)) < 1e-6 and compute_fgd(a, b) > 0.1


def test_trainer_epoch_loop_checkpoint_resume_and_test(tmp_path):
    """The caller around the hot path (trainer.py:367-427, 305-321, 172-224): Trainer.train over DataLoader batches for two
    epochs with validation, a checkpoint per epoch, LR schedule; resume from the first checkpoint reproduces the second
    epoch; Trainer.test evaluates a checkpoint."""
    import glob

    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.pipelines import get_pipeline

    def make(epochs):
        cfg = get_cfg_defaults()
        cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "voice2pose_sdt_bp.yaml"))
        cfg.merge_from_list(["DATASET.NAME", "SyntheticGestureDataset", "DATASET.SYNTHETIC_CLIPS", 8, "TRAIN.BATCH_SIZE", 4,
                             "TEST.BATCH_SIZE", 4, "TRAIN.NUM_EPOCHS", epochs, "SYS.NUM_WORKERS", 0, "SYS.LOG_INTERVAL", 1,
                             "SYS.OUTPUT_DIR", str(tmp_path), "TRAIN.SAVE_VIDEO", False, "TEST.SAVE_VIDEO", False,
                             "TEST.SAVE_NPZ", False])
        cfg.freeze()
        return get_pipeline(cfg.PIPELINE_TYPE)(cfg), cfg

    torch.manual_seed(11)
    pipe, cfg = make(2)
    pipe.train(cfg, "unit", None)  # the reference launcher's call pattern, main.py:51
    ckpts = sorted(glob.glob(os.path.join(str(tmp_path), "*unit", "checkpoints", "*.pth")))
    assert len(ckpts) == 2, ckpts
    final = {k: v.detach().clone() for k, v in pipe.model.state_dict().items()}
    c1 = torch.load(ckpts[0], map_location="cpu")
    assert c1["epoch"] == 1 and c1["step"] == 2 and all(k.startswith("module.") for k in c1["model_state_dict"])
    assert {"optimizerG_state_dict", "optimizerClipCode_state_dict"} <= set(c1)
    # resume from epoch 1 -> runs epoch 2 only.  The shuffled batch order of that epoch differs from the first run's (the
    # reference shuffles from the global RNG too), so the trained weights agree only to within the two Adam steps taken
    # (|dw| <= lr each); exact step-level resume equivalence is test_checkpoint_roundtrip_and_flat_buffers' job.
    pipe2, _ = make(2)
    pipe2.train(cfg, "unit", ckpts[0])
    start = {k[len("module."):]: v for k, v in c1["model_state_dict"].items()}
    lr = pipe2.optimizers["optimizerG"].param_groups[0]["lr"]
    # NUM_EPOCHS=2 puts the milestone E-2 = 0 at the very first scheduler step (torch semantics): lr = 1e-5 throughout,
    # and the resumed optimiser keeps the decayed lr it was saved with
    assert lr == pytest.approx(1e-5) and pipe.optimizers["optimizerG"].param_groups[0]["lr"] == pytest.approx(1e-5)
    moved = 0.0
    for k, v in pipe2.model.state_dict().items():
        if v.is_floating_point() and k.startswith("netG.") and "running" not in k:
            assert (v - final[k]).abs().max().item() <= 4.5 * lr, k
            moved = max(moved, (v.cpu() - start[k]).abs().max().item())
    assert 0.5 * lr < moved <= 2.5 * lr, moved  # it did train for exactly one more epoch (2 steps)
    assert len(glob.glob(os.path.join(os.path.dirname(ckpts[0]), "*.pth"))) == 2  # epoch-2 checkpoint rewritten in place
    # test mode from the final checkpoint
    pipe3, _ = make(2)
    out = pipe3.test(cfg, "unit_test", ckpts[1])  # main.py:48
    assert np.isfinite(float(out["G_loss"])) and "FGD_mu" in out


def test_pose2pose_epoch_loop_with_validation(tmp_path):
    """Config 5's caller path: Pose2Pose under Trainer.train with validation (pose2pose.py:124-217), TEST.MULTIPLE > 1."""
    import glob

    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.pipelines import get_pipeline
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "pose2pose.yaml"))
    cfg.merge_from_list(["DATASET.NAME", "SyntheticGestureDataset", "DATASET.SYNTHETIC_CLIPS", 8, "TRAIN.BATCH_SIZE", 4,
                         "TEST.BATCH_SIZE", 4, "TEST.MULTIPLE", 2, "TRAIN.NUM_EPOCHS", 1, "SYS.NUM_WORKERS", 0,
                         "SYS.LOG_INTERVAL", 1, "SYS.OUTPUT_DIR", str(tmp_path), "TRAIN.SAVE_VIDEO", False,
                         "TEST.SAVE_VIDEO", False, "TEST.SAVE_NPZ", True])
    cfg.freeze()
    pipe = get_pipeline(cfg.PIPELINE_TYPE)(cfg)
    pipe.train(cfg, "p2p", None)
    out = pipe.validate(pipe.test_dataloader, 1)  # trainer.py:407 signature
    for k in ("reg_loss", "kl_loss", "loss", "L2_dist", "lip_sync_error_n", "L2_dist_min", "L2_dist_max"):
        assert k in out and np.isfinite(float(out[k])), (k, out)
    assert float(out["L2_dist_min"]) <= float(out["L2_dist_max"])
    npz = glob.glob(os.path.join(str(tmp_path), "*p2p", "results", "epoch1-VAL-step*.npz"))  # the reference's file naming
    assert npz, "validation results were not saved"
    saved = np.load(npz[0])
    assert saved["poses_pred_batch"].shape == (8, 64, 2, 121) and saved["poses_pred_batch"].dtype == np.float64
    assert len(glob.glob(os.path.join(str(tmp_path), "*p2p", "checkpoints", "*.pth"))) == 1


def test_trainer_demo_loop_on_wav_files(tmp_path):
    """Row f-4: Trainer.demo (trainer.py:459-484) on wav input -- variable-length inference through the trained generator with
    the code interpolated between two training clips over DEMO.MULTIPLE passes (voice2pose.py:107-117,386-410)."""
    from scipy.io import wavfile

    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets.gesture_dataset import load_speaker_stats
    from speechdrivestemplates_amd.core.pipelines import get_pipeline
    load_speaker_stats(os.path.join(GOLDEN, "speaker_stat_oliver.npz"), "oliver")
    pipe, _ = _make_pipeline("voice2pose_sdt_bp", 16, 0.5)
    pipe.base_path = str(tmp_path)
    ckpt = pipe.save_checkpoint(1, 1)
    rng = np.random.default_rng(2)
    for name, secs in (("x.wav", 2.4), ("y.wav", 5.0)):
        wavfile.write(str(tmp_path / name), 16000, (rng.standard_normal(int(16000 * secs)) * 2000).astype(np.int16))
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "voice2pose_sdt_bp.yaml"))
    cfg.merge_from_list(["DATASET.SPEAKER", "oliver", "DEMO.MULTIPLE", 3, "DEMO.CODE_INDEX", 0, "DEMO.CODE_INDEX_B", 5,
                         "SYS.OUTPUT_DIR", str(tmp_path), "TEST.SAVE_NPZ", True, "TEST.SAVE_VIDEO", False])
    cfg.freeze()
    demo = get_pipeline(cfg.PIPELINE_TYPE)(cfg)
    outs = demo.demo(cfg, "demo", ckpt, "%s %s" % (tmp_path / "x.wav", tmp_path / "y.wav"))
    assert len(outs) == 2 * 3
    T = [36, 36, 36, 75, 75, 75]
    for o, t in zip(outs, T):
        p = o["poses_pred_batch"]
        assert p.shape == (1, t, 2, 121) and p.dtype == torch.float64 and torch.isfinite(p).all()
    table = demo.model.clips_code.detach()
    for i, coeff in enumerate((0.0, 0.5, 1.0)):  # the code is interpolated between rows 0 and 5 of the trained table
        want = table[0] * (1 - coeff) + table[5] * coeff
        assert torch.allclose(outs[i]["condition_code"][0], want, atol=1e-7)
    assert not torch.equal(outs[0]["poses_pred_batch"], outs[2]["poses_pred_batch"])


def test_pose2pose_demo_decodes_stored_codes(tmp_path):
    """Config 5's demo path (pose2pose.py:50-63,219-244): decode codes stored in DEMO.CODE_PATH; checked against the oracle's
    decoder on the same weights."""
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import gesture_dataset as gd
    from speechdrivestemplates_amd.core.pipelines import get_pipeline
    codes = np.random.default_rng(4).standard_normal((3, 32)).astype(np.float32) * 0.1
    np.savez(str(tmp_path / "codes.npz"), v=codes)
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "pose2pose.yaml"))
    cfg.merge_from_list(["DATASET.NAME", "SyntheticGestureDataset", "DATASET.SYNTHETIC_CLIPS", 8, "DEMO.MULTIPLE", 3,
                         "DEMO.CODE_PATH", str(tmp_path / "codes.npz"), "TEST.SAVE_NPZ", False])
    cfg.freeze()
    pipe = get_pipeline(cfg.PIPELINE_TYPE)(cfg)
    ocfg = O.cfg_named("pose2pose")
    st = O.make_pose2pose_state(ocfg, 8, seed=0)
    st["mel_transfm.spectrogram.window"], st["mel_transfm.mel_scale.fb"] = O.mel_window(), O.mel_filterbank()
    pipe.setup_model(cfg, state_dict={"module." + k: v for k, v in st.items()})
    pipe.test_dataset = gd.SyntheticGestureDataset(cfg=cfg, num_clips=8, split="val")
    batch = torch.utils.data.default_collate([pipe.test_dataset[0]])
    pipe.model.eval()
    for i, coeff in enumerate((0.0, 0.5, 1.0)):
        out = pipe.demo_step(batch, 1, extra_id=i, interpolation_coeff=coeff)
        assert out["poses_pred_batch"].shape == (1, 64, 2, 121) and out["poses_pred_batch"].dtype == torch.float64
        code = torch.from_numpy(codes[int(2 * coeff)] * 10).unsqueeze(0)
        assert torch.allclose(out["clip_code_mu"].cpu(), code)
        ref = O.pose_seq_decoder(st, "ae.decoder", code, ocfg, False).permute(0, 2, 1).reshape(1, 64, 2, 121)
        fin = pipe.test_dataset.get_final_results(ref, batch["speaker_stat"])
        check("pose2pose demo decode vs oracle", out["poses_pred_batch"], fin, 5e-4)


@pytest.mark.parametrize("opt", ["SAMPLE_FROM_NORMAL", "TEST_WITH_GT_CODE"])
def test_eval_time_code_sources(opt):
    """voice2pose.py:95-105: at test time the condition code can come from N(0,1) or from the pose encoder applied to the
    ground-truth poses instead of a random row of the trained table."""
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import gesture_dataset as gd
    from speechdrivestemplates_amd.core.pipelines import get_pipeline
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "voice2pose_sdt_bp.yaml"))
    cfg.merge_from_list(["DATASET.NAME", "SyntheticGestureDataset", "DATASET.SYNTHETIC_CLIPS", 8, "TEST.BATCH_SIZE", 4,
                         "TEST.SAVE_NPZ", False, "SYS.LOG_INTERVAL", 10 ** 9, "VOICE2POSE.GENERATOR.CLIP_CODE." + opt, True])
    cfg.freeze()
    pipe = get_pipeline(cfg.PIPELINE_TYPE)(cfg)
    pipe.num_train_samples = 8
    st = O.make_voice2pose_state(O.cfg_named("voice2pose_sdt_bp"), 8, seed=0, code_std=0.5)
    pipe.setup_model(cfg, state_dict={"module." + k: v for k, v in st.items()})
    pipe.test_dataset = gd.SyntheticGestureDataset(cfg=cfg, num_clips=8, split="val")
    pipe.model.eval()
    batch = torch.utils.data.default_collate([pipe.test_dataset[i] for i in range(4)])
    torch.manual_seed(5)
    with torch.no_grad():
        losses, res = pipe.model(batch, pipe.test_dataset)
    code = res["condition_code"]
    assert code.shape == (4, 32) and torch.isfinite(res["poses_pred_batch"]).all() and np.isfinite(float(losses["G_loss"]))
    table = pipe.model.clips_code.detach()
    assert not any(torch.equal(code[i], table[j]) for i in range(4) for j in range(8))  # not rows of the trained table
    if opt == "TEST_WITH_GT_CODE":
        mu_ref, _ = O.pose_seq_encoder({k: v.detach().cpu() for k, v in pipe.model.state_dict().items()}, "pose_encoder",
                                       batch["poses"], O.cfg_named("voice2pose_sdt_bp"), False)
        check("code = pose-encoder mean of the ground truth", code, mu_ref, 5e-4)


def ops_mod():
    from speechdrivestemplates_amd import ops
    return ops


def test_train_steps_repeat_bit_identically():
    """The default mode is run-to-run reproducible (the reference asks cuDNN for the same, main.py:37-38): two runs of three sdt_bp
    train steps from the same state and batches end with BIT-IDENTICAL generator weights and clip codes.  Weight gradients, the
    head's bias gradient and every partial tile of the stream-K kernels are summed in a fixed order; what is left of unordered
    arithmetic are float64 atomics over fp32-valued partial sums in the normalisation statistics (order-dependent below 2^-53
    relative, which the fp32 mean / rstd derived from them do not see)."""
    from speechdrivestemplates_amd import ops

    def run(det):
        prev = ops.DETERMINISTIC_DW, ops.USE_STREAMK_DW
        ops.DETERMINISTIC_DW, ops.USE_STREAMK_DW = det, det
        try:
            pipe, _ = _make_pipeline("voice2pose_sdt_bp", 16, 0.5)
            for step in range(3):
                losses, _ = pipe.forward_backward(O.make_batch(4, 16, step=step, seed=1))
                pipe.optimizer_updates(losses)
            torch.cuda.synchronize()
            return torch.cat([o.flat_param.detach().reshape(-1).clone() for o in pipe.optimizers.values()])
        finally:
            ops.DETERMINISTIC_DW, ops.USE_STREAMK_DW = prev
    assert ops.DETERMINISTIC_DW, "the ordered reductions are the default"
    runs = [run(True) for _ in range(3)]
    for k in (1, 2):
        diff = (runs[0] - runs[k]).abs().max().item()
        print("  default mode, run 0 vs run %d of 3 steps: bitwise equal %s, max |diff| %.3e" % (k, torch.equal(runs[0], runs[k]), diff))
        assert torch.equal(runs[0], runs[k])
    c, d = run(False), run(False)
    print("  fp32-atomic weight gradients (opt-out), two runs of 3 steps: max |diff| %.3e" % (c - d).abs().max().item())


@pytest.mark.parametrize("cfg_name,B", [("voice2pose_sdt_bp", 32), ("voice2pose_sdt_vae", 4), ("voice2pose_s2g", 4), ("pose2pose", 4)])
def test_train_steps_repeat_bit_identically_other_configs(cfg_name, B):
    """VERDICT r4 item 3 asked for the run-to-run test on the other configs as well (and on the full 32-clip step, where every stream-K launch hands
    partial tiles between workgroups): three runs of three train steps from the same state, batches and reparameterisation noise end with
    bit-identical weights.  What is unordered in these steps are fp64 atomics over fp32-valued partial sums (exact: their order cannot be seen)."""
    def run():
        torch.manual_seed(1234)  # pose2pose draws its reparameterisation noise from torch's generator
        pipe, _ = _make_pipeline(cfg_name, 64 if B == 32 else 16, 0.5 if cfg_name == "voice2pose_sdt_bp" else 0.0)
        for step in range(3):
            batch = O.make_batch(B, 64 if B == 32 else 16, step=step, seed=1)
            if cfg_name == "voice2pose_s2g":
                batch["speaker"] = ["oliver"] * B
            losses, _ = pipe.forward_backward(batch)
            pipe.optimizer_updates(losses)
        torch.cuda.synchronize()
        return torch.cat([o.flat_param.detach().reshape(-1).clone() for o in pipe.optimizers.values()])
    runs = [run() for _ in range(3)]
    for k in (1, 2):
        diff = (runs[0] - runs[k]).abs().max().item()
        print("  %s B=%d, run 0 vs run %d of 3 steps: bitwise equal %s, max |diff| %.3e" % (cfg_name, B, k, torch.equal(runs[0], runs[k]), diff))
        assert torch.equal(runs[0], runs[k])
