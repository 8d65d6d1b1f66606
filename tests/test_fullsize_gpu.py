"""Parity at BASELINE's full size (32 clips per GPU) against the CPU oracle itself, with float64-calibrated tolerances.

Round 1 tied the B=32 kernels together only through identities (adjoints, linearity, permutation) -- a consistently wrong
linear map passes all of them.  Here the HIP path is compared with the ORACLE on the same 32-clip batch:

  * one forward + backward of sdt_bp (learned clip codes + KL), sdt_vae (external codes) and pose2pose at B=32 against the
    oracle in float64 AND against the oracle's own fp32 run of the same step (1.5 s + a few seconds of CPU);
  * every conv layer shape at B=32: batch items 0 and 31 of forward / input-gradient against a float64 F.conv of that single
    item (no cross-sample terms), and the weight gradient of a cotangent that is non-zero on items {0, 31} only;
  * the bf16 product mode (BASELINE config 4) at B=32 with its own, stated, bf16 tolerances, and 'bf16x6' held to the fp32 bar;
  * the 3-step B=4 trajectories of every config against float64 runs of the REFERENCE (tests/golden/trajectories_B4_f64.npz).

How "as close to float64 as the fp32 reference" is measured
-----------------------------------------------------------
Forward quantities (losses, prediction) are continuous in the rounding noise: each one has to satisfy
        |HIP - f64| <= K_FWD * |ref_fp32 - f64| + floor.
Gradients are NOT: the network is piecewise linear (|.| loss, LeakyReLU), so two fp32 runs whose forward passes differ by 1e-6
take a different branch at a handful of elements, and ONE such event moves whole gradient tensors by 1e-3..1e-1 of their
max-norm.  tests/tools/p2p_flip.py shows it on the pose2pose B=4 fixture: the HIP prediction is as accurate as the fp32
reference's (6.6e-6 vs 5.1e-6 of float64) yet sign(pred - gt) differs at exactly 1 of 61952 elements, and that single sign is
the whole "100x worse" gradient gap (the head-bias gradient is off by exactly 2/N at one channel); at B=32 per-tensor ratios
|HIP-f64| / |ref32-f64| scatter from 0.03 to 1500 in BOTH directions (profiles/r02_parity_tables.txt).  Therefore:
  1. where the oracle runs next to the kernels (B=32) the float64 gradient is evaluated at the SAME L1 sign decisions as the
     fp32 run it is compared with (loss = mean(s * (pred - gt)), s = that run's own signs) -- this removes the loss-level events;
  2. the remaining events (LeakyReLU branches inside the network) hit individual tensors of either run at random -- over four
     batches (profiles/r02_seed_sweep.txt) the fp32 ORACLE's own median tensor error jumps between 8e-7 (no event) and 4.5e-4,
     its worst tensor between 3.7e-3 and 1.2e-2, and the worst tensors are different ones every time, for both implementations --
     so the comparison is between the two error DISTRIBUTIONS over the gradient tensors of a step, anchored on the reference's
     worst tensor E_ref = max_k e_ref[k] (its only statistic that is stable from batch to batch):
        max_k e_hip[k]    <= K_MAX * E_ref
        median_k e_hip[k] <= max(K_MED * median_k e_ref[k], E_ref / 20)
        ||g_hip - g_f64||_2 / ||g_f64||_2 <= max(K_L2 * (same for ref32), E_ref / 20)      (all tensors concatenated)
     with e[k] = max|g[k] - g_f64[k]| / max|g_f64[k]|: the typical HIP gradient tensor is closer to float64 than a twentieth of the
     reference's own worst one (round 2: a quarter), and no HIP tensor is further than 3x that worst one.  Round 3, with the
     chunked accumulation of the stream-K kernels: sdt_bp B=32 median 7.5e-4 vs 4.5e-4 (1.7x; was 1.3e-3), whole-gradient L2 1.4x
     (was 3.2x) -- both now inside the plain 2x bar; sdt_vae B=32 still shows "HIP has events, the reference has none on this
     batch" (median 2.0e-4 vs 5.9e-7) while its whole-gradient L2 error is BELOW the reference's (2.95e-4 vs 3.87e-4);
  3. on the B=4 fixtures (no oracle in the loop, float64 gradients stored at the float64 run's own signs) L1 sign decisions
     that differ from the float64 run are COUNTED from the stored full predictions and each one is allowed its measured
     worst-case effect (FLIP_ALLOW of a tensor's max-norm); with zero differing decisions the bar is the one of item 2.
Every run appends its full per-tensor table to $SDT_PARITY_TABLES (committed as profiles/r02_parity_tables.txt).
"""
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import sdt_oracle as O
from test_model_gpu import _make_pipeline, sl

pytestmark = pytest.mark.gpu
DEV = "cuda"
K_FWD = 2.0   # per forward quantity: HIP at most this many times further from float64 than the fp32 reference (round 2: 3.0)
K_MED, K_MAX, K_L2 = 2.0, 3.0, 2.0  # gradient error distributions over the tensors of a step (see module docstring)
FLIP_ALLOW = 0.1  # B=4 fixtures: measured worst effect of ONE differing L1 sign on a gradient tensor (8.8e-2 of max, head weight)
N_CLIPS = 64


def _relmax(a, ref):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a, dtype=np.float64))
    ref = ref.detach().double().cpu() if torch.is_tensor(ref) else torch.as_tensor(np.asarray(ref, dtype=np.float64))
    assert a.shape == ref.shape, (a.shape, ref.shape)
    assert torch.isfinite(a).all()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def _oracle_grads(cfg_name, state32, batch32, dtype, eps32=None, l1_signs=None):
    """One forward + backward of the oracle on copies of ``state32`` / ``batch32`` cast to ``dtype`` (float64 runs see exactly
    the fp32 weights and inputs the other runs see).  ``l1_signs``: evaluate the regression term at these fixed sign decisions,
    mean(s * (pred - gt)) -- equal to mean|pred - gt| wherever s = sign(pred - gt).  Returns (losses, prediction, {name: grad})."""
    cfg = O.cfg_named(cfg_name)
    st = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state32.items()}
    batch = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch32.items()}
    if cfg_name == "pose2pose":
        O.OraclePose2Pose(cfg, st)
        losses, res = O.pose2pose_forward(st, batch, cfg, eps32.to(dtype), True)
        total, reg_key, lam, pref = "loss", "reg_loss", cfg.POSE2POSE.LAMBDA_REG, "ae."
    else:
        O.OracleVoice2Pose(cfg, st)
        losses, res = O.voice2pose_forward(st, batch, cfg, True)
        total, reg_key, lam, pref = "G_loss", "G_reg_loss", cfg.VOICE2POSE.GENERATOR.LAMBDA_REG, ("netG.", "clips_code")
    loss = losses[total]
    if l1_signs is not None:
        lin = (l1_signs.to(dtype) * (res["poses_pred_batch"] - batch["poses"])).mean() * lam
        loss = loss - losses[reg_key] + lin
    loss.backward()
    grads = {k: v.grad.detach() for k, v in st.items() if v.requires_grad and v.grad is not None and k.startswith(pref)}
    return {k: v.detach() for k, v in losses.items()}, res["poses_pred_batch"].detach(), grads


def _dump(lines):
    print("\n".join("  " + ln for ln in lines))
    dump = os.environ.get("SDT_PARITY_TABLES")
    if dump:
        with open(dump, "a") as f:
            f.write("\n".join(lines) + "\n")


def _check_forward(title, rows):
    """rows: (name, hip, ref32, f64, floor_rel)."""
    lines, bad = ["%s -- forward quantities, per quantity |hip-f64| <= %.0f x |ref32-f64| + floor:" % (title, K_FWD)], []
    for name, hip, r32, f64, floor in rows:
        eh, er = _relmax(hip, f64), _relmax(r32, f64)
        ok = eh <= K_FWD * er + floor
        lines.append("      %-44s hip %.3e  ref32 %.3e  ratio %6.2f%s" % (name, eh, er, eh / max(er, 1e-30), "" if ok else "  FAIL"))
        if not ok:
            bad.append((name, eh, er))
    _dump(lines)
    assert not bad, (title, bad)


def _check_grad_distributions(title, e_hip, e_ref, l2_hip=None, l2_ref=None, allow=0.0, k_med=K_MED, k_max=K_MAX, anchor=0.05):
    """e_hip / e_ref: {tensor name: max|g - g_f64| / max|g_f64|}.  ``allow``: extra absolute allowance on every statistic (B=4
    fixtures: FLIP_ALLOW per counted differing L1 sign decision)."""
    K_MED, K_MAX = k_med, k_max  # noqa: N806 (shadow the module defaults for this call)
    names = sorted(e_ref)
    eh, er = np.array([e_hip[k] for k in names]), np.array([e_ref[k] for k in names])
    med_h, med_r, max_h, max_r = np.median(eh), np.median(er), eh.max(), er.max()
    lines = ["%s -- gradient error distributions over %d tensors (allowance %.1e):" % (title, len(names), allow),
             "      median  hip %.3e  ref32 %.3e  (bar max(%.0fx, E_ref/%.0f))      max  hip %.3e  E_ref = ref32 %.3e  (bar %.0fx)"
             % (med_h, med_r, K_MED, 1.0 / anchor, max_h, max_r, K_MAX)]
    ok = med_h <= max(K_MED * med_r, max_r * anchor) + allow and max_h <= K_MAX * max_r + allow
    if l2_hip is not None:
        lines.append("      whole-gradient relative L2 error  hip %.3e  ref32 %.3e  (bar max(%.0fx, E_ref/%.0f))" % (l2_hip, l2_ref, K_L2, 1.0 / anchor))
        ok = ok and l2_hip <= max(K_L2 * l2_ref, max_r * anchor) + allow
    for k, a, b in zip(names, eh, er):
        lines.append("      %-60s hip %.3e  ref32 %.3e  ratio %8.2f" % (k, a, b, a / max(b, 1e-30)))
    _dump(lines)
    assert ok, (title, "median", med_h, med_r, "max", max_h, max_r, "L2", l2_hip, l2_ref)


def _flat_l2(ga, gb, names):
    num = sum(float(((ga[k].double().cpu() - gb[k].double().cpu()) ** 2).sum()) for k in names)
    den = sum(float((gb[k].double().cpu() ** 2).sum()) for k in names)
    return (num / den) ** 0.5


def _b32_run(cfg_name, code_std, conv_math="f32"):
    """HIP forward+backward at B=32 in ``conv_math`` plus the oracle in fp32 and in float64 (the latter twice: at the HIP run's
    and at the fp32 oracle's L1 sign decisions)."""
    from speechdrivestemplates_amd import ops
    B = 32
    ocfg = O.cfg_named(cfg_name)
    batch = O.make_batch(B, N_CLIPS, step=3, seed=11)
    eps = None
    ops.set_conv_math(conv_math)
    try:
        pipe, _ = _make_pipeline(cfg_name, N_CLIPS, code_std)
        if cfg_name == "pose2pose":
            state = O.make_pose2pose_state(ocfg, N_CLIPS, seed=0)
            eps = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal((B, 32)).astype(np.float32))
            real_randn = torch.randn
            torch.randn = lambda *a, **k: eps.clone().to(DEV)
            try:
                losses, results = pipe.forward_backward(batch)
            finally:
                torch.randn = real_randn
            pred_hip, loss_keys = results["poses_pred_batch"], ("reg_loss", "kl_loss", "loss")
        else:
            state = O.make_voice2pose_state(ocfg, N_CLIPS, seed=0, code_std=code_std)
            if ocfg.VOICE2POSE.GENERATOR.CLIP_CODE.EXTERNAL_CODE:
                state["clips_code"] = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal((N_CLIPS, 32)).astype(np.float32))
            losses, results = pipe.forward_backward(batch)
            pred_hip, loss_keys = results["poses_pred_normalized"], ("G_reg_loss", "G_clipcode_kl_loss", "G_loss")
        torch.cuda.synchronize()
    finally:
        ops.set_conv_math("f32")
    grads_hip = {k: p.grad.detach().clone() for k, p in pipe.model.named_parameters() if p.grad is not None}
    gt = batch["poses"]
    t0 = time.time()
    l32, p32, g32 = _oracle_grads(cfg_name, state, batch, torch.float32, eps)
    t1 = time.time()
    s_hip = torch.sign(pred_hip.detach().cpu() - gt)
    s_ref = torch.sign(p32 - gt)
    l64, p64, g64_hip = _oracle_grads(cfg_name, state, batch, torch.float64, eps, l1_signs=s_hip)
    _, _, g64_ref = _oracle_grads(cfg_name, state, batch, torch.float64, eps, l1_signs=s_ref)
    n_flip_hip = int((s_hip != torch.sign(p64 - gt.double())).sum())
    n_flip_ref = int((s_ref != torch.sign(p64 - gt.double())).sum())
    print("  oracle B=32 forward+backward: fp32 %.1f s, float64 2 x %.1f s; L1 sign decisions differing from float64: hip %d, ref32 %d of %d"
          % (t1 - t0, (time.time() - t1) / 2, n_flip_hip, n_flip_ref, gt.numel()))
    return dict(losses=losses, loss_keys=loss_keys, pred_hip=pred_hip, grads_hip=grads_hip, l32=l32, p32=p32, g32=g32, l64=l64, p64=p64,
                g64_hip=g64_hip, g64_ref=g64_ref, batch=batch)


def _b32_check(title, r, metrics=True):
    _check_forward(title, [("loss " + k, r["losses"][k].detach().reshape(1), r["l32"][k].reshape(1), r["l64"][k].reshape(1), 3e-7)
                           for k in r["loss_keys"]] + [("prediction (32,64,2,121)", r["pred_hip"], r["p32"], r["p64"], 2e-7)])
    names = sorted(r["g64_hip"])
    assert set(names) <= set(r["grads_hip"]), sorted(set(names) - set(r["grads_hip"]))[:5]
    e_hip = {k: _relmax(r["grads_hip"][k], r["g64_hip"][k]) for k in names}
    e_ref = {k: _relmax(r["g32"][k], r["g64_ref"][k]) for k in names}
    _check_grad_distributions(title, e_hip, e_ref, _flat_l2(r["grads_hip"], r["g64_hip"], names), _flat_l2(r["g32"], r["g64_ref"], names))
    if metrics:  # float64 metrics of the step
        fin_p = O.get_final_results(r["p64"].clone(), r["batch"]["speaker_stat"], True)
        fin_g = O.get_final_results(r["batch"]["poses"].double(), r["batch"]["speaker_stat"], True)
        m64 = O.evaluate_step(fin_p, fin_g)
        assert abs(float(r["losses"]["L2_dist"]) - float(m64["L2_dist"])) <= 1e-5 * float(m64["L2_dist"])
        assert abs(float(r["losses"]["lip_sync_error_n"]) - float(m64["lip_sync_error_n"])) <= 2e-4 * float(m64["lip_sync_error_n"])


@pytest.mark.parametrize("cfg_name,code_std", [("voice2pose_sdt_bp", 0.5), ("voice2pose_sdt_vae", 0.0), ("pose2pose", 0.0)])
def test_b32_forward_backward_vs_oracle_f64_calibrated(cfg_name, code_std):
    _b32_check("%s B=32 (fp32 MFMA) vs float64 oracle" % cfg_name, _b32_run(cfg_name, code_std))


def test_b32_bf16_mode_vs_oracle():
    """BASELINE config 4 (sdt_bp, bf16): conv products from bf16-rounded operands, fp32 accumulation and fp32 everywhere else.
    Stated bf16 tolerances at 32 clips per GPU, against the float64 oracle (evaluated at the run's own L1 sign decisions):
    prediction 4e-2 of max, losses 2e-2, every gradient tensor within 40 % of its max-norm and with cosine similarity >= 0.97 to
    the float64 gradient (bf16 has 8 significand bits: 2^-9 = 2e-3 per product, amplified through 25 normalised layers).  The max-norm
    bar is set by ONE tensor, the first encoder conv (64 x 1 x 3 x 3: the end of the longest chain, a single largest element decides the
    ratio): measured 0.24-0.27 from run to run (the statistics' fp64 atomics are unordered), cosine 0.98; the errors fall off along the chain
    (encoder convs 0.27, 0.20, 0.21, 0.20, 0.17, 0.10, 0.09, 0.07; U-Net / decoder below that: profiles/r04_parity_tables.txt).
    The bf16-STORAGE test of the same config states the same 40 % (tests/test_bf16_gpu.py)."""
    from speechdrivestemplates_amd import ops
    B, cfg_name = 32, "voice2pose_sdt_bp"
    ocfg = O.cfg_named(cfg_name)
    state = O.make_voice2pose_state(ocfg, N_CLIPS, seed=0, code_std=0.5)
    batch = O.make_batch(B, N_CLIPS, step=3, seed=11)
    ops.set_conv_math("bf16")
    try:
        pipe, _ = _make_pipeline(cfg_name, N_CLIPS, 0.5)
        losses, results = pipe.forward_backward(batch)
        torch.cuda.synchronize()
    finally:
        ops.set_conv_math("f32")
    grads_hip = {k: p.grad.detach().double().cpu() for k, p in pipe.model.named_parameters() if p.grad is not None}
    s_hip = torch.sign(results["poses_pred_normalized"].detach().cpu() - batch["poses"])
    l64, p64, g64 = _oracle_grads(cfg_name, state, batch, torch.float64, l1_signs=s_hip)
    e = _relmax(results["poses_pred_normalized"], p64)
    rows = []
    for k, ref in g64.items():
        got = grads_hip[k]
        rel = ((got - ref).abs().max() / ref.abs().max()).item()
        cos = (F.cosine_similarity(got.reshape(1, -1), ref.reshape(1, -1)).item()) if ref.numel() > 1 else 1.0
        rows.append((k, rel, cos))
    lines = ["voice2pose_sdt_bp B=32 (bf16 products) vs float64 oracle: prediction rel-max-err %.3e; losses %s" % (
        e, {k: "%.2e" % abs(float(losses[k].detach()) / float(l64[k]) - 1.0) for k in ("G_reg_loss", "G_clipcode_kl_loss", "G_loss")})]
    lines += ["      %-60s rel-max-err %.3e  cosine %.5f" % r for r in rows]
    _dump(lines)
    assert e <= 4e-2, e
    for k in ("G_reg_loss", "G_clipcode_kl_loss", "G_loss"):
        a, b = float(losses[k]), float(l64[k])
        assert abs(a - b) <= 2e-2 * abs(b), (k, a, b)
    worst_rel, worst_cos = max(r[1] for r in rows), min(r[2] for r in rows)
    print("  bf16 B=32: worst gradient rel-max-err %.3e, worst cosine %.5f" % (worst_rel, worst_cos))
    assert worst_rel <= 0.40 and worst_cos >= 0.97, [r for r in rows if r[1] > 0.40 or r[2] < 0.97]


def test_b32_bf16x6_mode_meets_the_fp32_bar():
    """'bf16x6' (operands split exactly into three bf16 pieces inside the conv kernels, six MFMA products, fp32 accumulation) claims
    fp32-equivalent products: at B=32 it has to pass the SAME float64-calibrated checks as the exact-fp32 MFMA path."""
    r = _b32_run("voice2pose_sdt_bp", 0.5, conv_math="bf16x6")
    _b32_check("voice2pose_sdt_bp B=32 (bf16x6 products, split in kernel) vs float64 oracle", r)


# ----------------------------------------------------------------------------------------------------------------------
# every conv layer shape at B=32: single batch items against float64
# ----------------------------------------------------------------------------------------------------------------------
CONV_B32 = [  # name, Hi, Wi, Cin, Cout, kh, kw, s, p  (Hi == 1: Conv1d)
    ("L1", 80, 427, 64, 64, 4, 4, 2, 1), ("L2", 40, 213, 64, 128, 3, 3, 1, 1), ("L3", 40, 213, 128, 128, 4, 4, 2, 1),
    ("L4", 20, 106, 128, 256, 3, 3, 1, 1), ("L5", 20, 106, 256, 256, 4, 4, 2, 1), ("L6", 10, 53, 256, 256, 3, 3, 1, 1),
    ("L7", 10, 53, 256, 256, 6, 3, 1, 0),
    ("unet e0 288->256 k3 T64", 1, 64, 288, 256, 1, 3, 1, 1), ("unet k4s2 T64", 1, 64, 256, 256, 1, 4, 2, 1),
    ("unet k4s2 T4", 1, 4, 256, 256, 1, 4, 2, 1), ("unet k3 T8", 1, 8, 256, 256, 1, 3, 1, 1),
    ("head 256->242 k1", 1, 64, 256, 242, 1, 1, 1, 0), ("pose-enc 242->256 k3", 1, 64, 242, 256, 1, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_B32, ids=[c[0] for c in CONV_B32])
def test_conv_b32_single_items_vs_float64(case):
    from speechdrivestemplates_amd import ops
    tag, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    B, items = 32, (0, 31)
    one_d = Hi == 1
    gen = torch.Generator().manual_seed(1000 + sum(map(ord, tag)))
    xs, ws = ((B, Wi, Cin), (Cout, Cin, kw)) if one_d else ((B, Hi, Wi, Cin), (Cout, Cin, kh, kw))
    x = torch.randn(xs, generator=gen)
    w = torch.randn(ws, generator=gen) * (2.0 / (Cin * kh * kw)) ** 0.5
    xd = x.to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w).to(DEV))
    yd = ops.conv_forward(xd, wd, None, s, p)
    gy = torch.randn(yd.shape, generator=gen)
    gyd = gy.to(DEV)
    dxd = ops.conv_input_grad(gyd, wd, xd.shape, s, p)
    mask = torch.zeros(B, *([1] * (gy.dim() - 1)))
    mask[list(items)] = 1.0
    ops.conv_weight_grad(xd, (gy * mask).to(DEV).contiguous(), wd, s, p)
    torch.cuda.synchronize()
    if not one_d:
        # these launches must be the HEADLINE's kernels -- the split-fp32 forward / input-gradient plan (bit 26 of the plan header)
        # and the split weight-gradient plan -- not a fallback that happens to be accurate too
        assert ops.F32_SPLIT and ops.USE_STREAMK and ops.USE_STREAMK_DW
        g = ops.conv_geom_for(xd.shape, wd, s, p)
        fplan = ops._sk_plan(g, 1, 0, 1, xd.device, forward=True)
        assert fplan is not None and (fplan.host[3] >> 26) & 1, (tag, "forward did not take the split-fp32 kernel")
        arr, n, _gs = ops.dx_pack(B, Hi, Wi, Cin, Cout, kh, kw, s, p, False)
        xplan = ops._sk_plan(arr, n, -1, 1, xd.device)
        assert xplan is not None and (xplan.host[3] >> 26) & 1, (tag, "input gradient did not take the split-fp32 kernel")
        wplan = ops._sk_dw_plan(g, xd.device)
        assert wplan is not None and (wplan.host[3] >> 26) & 1, (tag, "weight gradient did not take convx3_dw_kernel")
    conv = F.conv1d if one_d else F.conv2d
    dw_ref = torch.zeros_like(w, dtype=torch.float64)
    for b in items:
        xb = (x[b:b + 1].permute(0, 2, 1) if one_d else x[b:b + 1].permute(0, 3, 1, 2)).double().requires_grad_(True)
        wb = w.double().requires_grad_(True)
        yb = conv(xb, wb, None, s, p)
        gb = (gy[b:b + 1].permute(0, 2, 1) if one_d else gy[b:b + 1].permute(0, 3, 1, 2)).double()
        yb.backward(gb)
        y_cl = yb.detach().permute(0, 2, 1) if one_d else yb.detach().permute(0, 2, 3, 1)
        dx_cl = xb.grad.permute(0, 2, 1) if one_d else xb.grad.permute(0, 2, 3, 1)
        assert _relmax(yd[b:b + 1], y_cl) < 3e-6, (tag, "fwd item", b, _relmax(yd[b:b + 1], y_cl))
        assert _relmax(dxd[b:b + 1], dx_cl) < 3e-6, (tag, "dX item", b, _relmax(dxd[b:b + 1], dx_cl))
        dw_ref += wb.grad
    e = _relmax(wd.grad, dw_ref)
    print("  %-28s B=32 items %s: dW (masked cotangent) rel-max-err %.2e" % (tag, items, e))
    assert e < 5e-6, (tag, "dW", e)


# ----------------------------------------------------------------------------------------------------------------------
# 3-step B=4 trajectories of every config: fp64-calibrated per step and per gradient tensor
# ----------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_f64():
    return dict(np.load(os.path.join(GOLDEN, "trajectories_B4_f64.npz")))


def _sample_err(got, f64):
    got, f64 = np.asarray(got, dtype=np.float64), np.asarray(f64, dtype=np.float64)
    return float(np.abs(got - f64).max() / max(np.abs(f64).max(), 1e-30))


@pytest.mark.parametrize("name,code_std", [("voice2pose_sdt_bp", 0.5), ("voice2pose_sdt_bp_zero", 0.0), ("voice2pose_s2g", 0.0),
                                           ("voice2pose_sdt_vae", 0.0), ("pose2pose", 0.0)])
def test_trajectory_is_as_close_to_float64_as_the_fp32_reference(golden_traj, golden_f64, name, code_std):
    """Replaces round 1's flat gradient tolerance (2e-2 of max on 64 samples, whatever the tensor): every step of the 3-step run
    is compared with the REFERENCE's float64 run of the same trajectory.
      step 0 (identical weights on all three sides): losses and prediction per quantity within K_FWD of the reference's own
        fp32-vs-float64 distance; the 64 stored samples of every gradient tensor through the distribution check of the module
        docstring, with FLIP_ALLOW for every L1 sign decision that differs from the float64 run (counted from the stored full
        predictions; the reference's fp32 run is given the same accounting);
      steps 1-2: the three runs have taken different-but-equivalent Adam steps (|dw| = lr * sign(g) on the first steps, so fp32
        noise on near-zero gradients flips update signs): the gradient error distributions only have to overlap -- worst HIP
        tensor within 5x the reference's worst, median within max(5x the reference's median, the reference's worst)."""
    cfg_name = name.replace("_zero", "")
    pipe, cfg = _make_pipeline(cfg_name, 16, code_std)
    g32 = {k[len(name) + 1:]: v for k, v in golden_traj.items() if k.startswith(name + "/")}
    g64 = {k[len(name) + 1:]: v for k, v in golden_f64.items() if k.startswith(name + "/")}
    real_randn = torch.randn
    other = _OtherConvArithmetic(cfg_name, code_std) if cfg_name != "pose2pose" else None
    for step in range(3):
        batch = O.make_batch(4, 16, step=step, seed=1)
        if cfg_name == "voice2pose_s2g":
            batch["speaker"] = ["oliver"] * 4
        if cfg_name == "pose2pose":
            eps = torch.from_numpy(np.random.Generator(np.random.PCG64([2, step])).standard_normal((4, 32)).astype(np.float32)).to(DEV)
            torch.randn = lambda *a, **k: eps.clone()
        dec = _Decisions()
        try:
            with dec:
                losses, results = pipe.forward_backward(batch)
        finally:
            torch.randn = real_randn
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in pipe.model.named_parameters() if p.grad is not None}
        pred = (results["poses_pred_normalized"] if "poses_pred_normalized" in results else results["poses_pred_batch"]).detach().cpu()
        title = "%s B=4 step %d vs float64 reference" % (name, step)
        n_flip = 0
        if step == 0:
            full32 = g32["s0/pred_full"] if "s0/pred_full" in g32 else g32["s0/pred"]
            full64 = g64["s0/pred_full"] if "s0/pred_full" in g64 else g64["s0/pred"]
            rows = [("loss " + k.split("/")[-1], [float(losses[k.split("/")[-1]])], [float(g32[k])], [float(g64[k])], 1e-6)
                    for k in g32 if k.startswith("s0/loss/") and k in g64 and k.split("/")[-1] in losses]
            rows.append(("prediction", pred.numpy(), full32, full64, 2e-7))
            _check_forward(title, rows)
            gt = batch["poses"].double().numpy()
            s64 = np.sign(full64 - gt)
            n_flip = int((np.sign(pred.double().numpy() - gt) != s64).sum())
            n_flip_ref = int((np.sign(full32.astype(np.float64) - gt) != s64).sum())
            print("  %s: L1 sign decisions differing from the float64 run: hip %d, reference fp32 %d (of %d)" % (title, n_flip, n_flip_ref, gt.size))
        keys = [k for k in g64 if k.startswith("s%d/grad/" % step) and not k.startswith("s%d/grad/Dstep:" % step)
                and k in g32 and k.split("/grad/")[1] in grads]
        e_hip = {k.split("/grad/")[1]: _sample_err(sl(grads[k.split("/grad/")[1]])[:64], g64[k][:64]) for k in keys}
        e_ref = {k.split("/grad/")[1]: _sample_err(g32[k][:64], g64[k][:64]) for k in keys}
        g64_of = lambda k, _s=step: g64["s%d/grad/%s" % (_s, k)][:64]  # noqa: E731
        if step == 0:
            _check_or_event(other, step, False, title, e_hip, e_ref, g64_of, decisions=dec.rec, allow=FLIP_ALLOW * n_flip)
        else:  # behind >= 1 Adam update: chaotic sign noise on both sides -- the distributions only have to overlap
            _check_or_event(other, step, False, title, e_hip, e_ref, g64_of, decisions=dec.rec, k_med=5.0, k_max=5.0, anchor=1.0)
        with dec:
            pipe.optimizer_updates(losses)
        if cfg_name == "voice2pose_s2g":  # second backward (discriminator step): its gradients exist after optimizer_updates
            torch.cuda.synchronize()
            params = dict(pipe.model.named_parameters())
            keys = [k for k in g64 if k.startswith("s%d/grad/Dstep:" % step) and k in g32]
            e_hip = {k.split("Dstep:")[1]: _sample_err(sl(params[k.split("Dstep:")[1]].grad)[:64], g64[k][:64]) for k in keys}
            e_ref = {k.split("Dstep:")[1]: _sample_err(g32[k][:64], g64[k][:64]) for k in keys}
            _check_or_event(other, step, True, title + " (discriminator step)", e_hip, e_ref,
                            lambda k, _s=step: g64["s%d/grad/Dstep:%s" % (_s, k)][:64], decisions=dec.rec, k_med=5.0, k_max=5.0, anchor=1.0)


class _Decisions:
    """Records the LeakyReLU decisions of everything that runs inside the ``with`` block: the sign bit of every activated output of the norm+activation
    ops (ops.ColNormActFn / RowNormActFn / L0BlockFn / ConvRowNormFn) and -- the Conv1d chain -- of every block input it keeps for the backward pass
    plus its output.  ``rec`` is a list of bool tensors in call order; two runs of the same step through the same routing give lists of equal layout."""
    FNS = ("ColNormActFn", "RowNormActFn", "L0BlockFn", "ConvRowNormFn", "Chain1dFn")

    def __init__(self):
        self.rec, self._orig = [], {}

    def __enter__(self):
        from speechdrivestemplates_amd import ops
        for name in self.FNS:
            cls = getattr(ops, name)
            orig = self._orig[name] = cls.__dict__["forward"]
            fn = orig.__func__ if isinstance(orig, staticmethod) else orig

            def wrapped(ctx, *a, _fn=fn, _name=name):
                out = _fn(ctx, *a)
                z = out[0] if isinstance(out, tuple) else out
                self.rec.append(torch.signbit(z.detach()))
                if _name == "Chain1dFn":
                    self.rec.extend(torch.signbit(x.detach()) for x in (getattr(ctx, "xs", None) or []) if x is not None)
                return out

            setattr(cls, "forward", staticmethod(wrapped))
        return self

    def __exit__(self, *exc):
        from speechdrivestemplates_amd import ops
        for name, orig in self._orig.items():
            setattr(getattr(ops, name), "forward", orig)
        return False

    @staticmethod
    def differing(a, b):
        """number of activation decisions that differ between two recordings of the same step; ``a`` may stop earlier than ``b`` (recorded up to the
        generator's backward pass only): the common prefix is compared, and its layout must agree"""
        assert 0 < len(a) <= len(b) and all(x.shape == y.shape for x, y in zip(a, b)), ("the two runs took different routes", len(a), len(b))
        return int(sum(int((x != y).sum().item()) for x, y in zip(a, b)))


ACT_EVENT_ALLOW = 1e-2  # B=4 fixtures: measured worst effect of ONE differing LeakyReLU decision on a gradient tensor (8.3e-3 of max: unet.e3.norm.weight
#                          behind a flip in the pose discriminator's second activation, 8192 elements)


class _OtherConvArithmetic:
    """Activation-sign EVENTS on a fixed fixture batch (module docstring, item 2): a LeakyReLU input within fp32 noise of zero lands on the other side
    than in the float64 run, and every gradient tensor upstream of it moves by ~1 / sqrt(elements of that activation) -- 4e-3 .. 8e-3 for the pose
    discriminator at B = 4, far outside bars that are anchored on an event-free reference run.  WHICH run has such an event on a given batch is a
    property of its rounding, not of its accuracy.  The product has two fp32-grade arithmetic variants of the Conv2d forward / input gradient
    (ops.F32_SPLIT: split-fp32 products, default; fp32 MFMA, rounds 3-4) whose forward results differ in the last bits only.  When a check of the
    default variant fails, the SAME steps are replayed with the other variant: that one must meet the unchanged bars (a gradient bug common to both
    fails here; the split kernel's own results are held to 3e-6 of float64 per layer by test_conv_b32_single_items_vs_float64 and
    test_split_f32_conv_vs_float64_and_the_fp32_mfma_kernels), and the failing variant is then held to the same bars plus ONE event's measured effect."""

    def __init__(self, cfg_name, code_std):
        self.cfg_name, self.code_std = cfg_name, code_std
        self.pipe, self.steps = None, []  # per replayed step: (generator-step grads, discriminator-step grads, activation decisions of the step)

    def grads(self, step, dstep=False):
        from speechdrivestemplates_amd import ops
        prev = ops.F32_SPLIT
        ops.F32_SPLIT = not prev
        try:
            if self.pipe is None:
                self.pipe, _ = _make_pipeline(self.cfg_name, 16, self.code_std)
            while len(self.steps) <= step:
                batch = O.make_batch(4, 16, step=len(self.steps), seed=1)
                if self.cfg_name == "voice2pose_s2g":
                    batch["speaker"] = ["oliver"] * 4
                with _Decisions() as dec:
                    losses, _ = self.pipe.forward_backward(batch)
                    torch.cuda.synchronize()
                    g = {k: p.grad.detach().clone() for k, p in self.pipe.model.named_parameters() if p.grad is not None}
                    self.pipe.optimizer_updates(losses)
                    torch.cuda.synchronize()
                d = {k: p.grad.detach().clone() for k, p in self.pipe.model.named_parameters() if p.grad is not None}
                self.steps.append((g, d, dec.rec))
        finally:
            ops.F32_SPLIT = prev
        return self.steps[step][1 if dstep else 0]

    def decisions(self, step):
        self.grads(step)
        return self.steps[step][2]


def _check_or_event(other, step, dstep, title, e_hip, e_ref, g64_of, decisions=None, **kw):
    """The unchanged bars; on failure the event protocol of _OtherConvArithmetic (voice2pose configs only: pose2pose has no Conv2d).
    ``decisions``: the default variant's recorded activation decisions of this step (_Decisions.rec).  The allowance is granted only when the EVENT
    IS PROVEN: at least one LeakyReLU decision of this step differs between the two arithmetic variants -- a gradient defect of the
    default kernels that leaves every decision alone gets no allowance and fails the unchanged bars."""
    try:
        _check_grad_distributions(title, e_hip, e_ref, **kw)
        return
    except AssertionError:
        if other is None or decisions is None:
            raise
    from speechdrivestemplates_amd import ops
    g2 = other.grads(step, dstep)
    n_diff = _Decisions.differing(decisions, other.decisions(step))
    print("  %s: activation decisions differing between F32_SPLIT = %s and %s: %d" % (title, ops.F32_SPLIT, not ops.F32_SPLIT, n_diff))
    assert n_diff >= 1, (title, "the default arithmetic fails the unchanged bars and no activation decision differs from the other variant: not an event")
    e_other = {k: _sample_err(sl(g2[k])[:64], g64_of(k)) for k in e_hip}
    _check_grad_distributions(title + " [the other fp32 conv arithmetic, F32_SPLIT = %s: unchanged bars]" % (not ops.F32_SPLIT), e_other, e_ref, **kw)
    kw = dict(kw, allow=kw.get("allow", 0.0) + ACT_EVENT_ALLOW)
    _check_grad_distributions(title + " [activation-sign event under F32_SPLIT = %s: + one event's allowance]" % ops.F32_SPLIT, e_hip, e_ref, **kw)


def _stage_err(a, ref):
    """(max|a-ref| / max|ref|, rms(a-ref) / rms(ref)) against a float64 reference"""
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    d = a - ref
    return (d.abs().max() / ref.abs().max()).item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def test_b32_forward_stage_error_table():
    """Where does the forward error come from?  (the HIP prediction was 2.6x further from float64 than the fp32
    reference's and nobody had localised it.)  Every stage of the sdt_bp generator runs ALONE on the same fp32 input -- the float64
    chain's activations rounded to fp32 -- on the HIP path and on the fp32 oracle, and both are compared with the float64 stage on
    that input; then the same for the cumulative chain (each implementation feeding itself).  Errors as max-norm and as RMS.
    The bar: no isolated stage more than K_STAGE = 2 x the fp32 reference's RMS error (+ a floor of 2e-8: one stage of the
    reference can be exact by luck; the DFT-as-GEMM mel against the reference's FFT: 3 x), the whole chain within K_FWD.
    History (profiles/r03_stage_errors_before.txt): with one K-long accumulator per output the Conv2d layers sat at 1.9-3.8 x, growing
    with sqrt(K); the stream-K kernel accumulates in chunks of 256 products and is level with the reference (1.0-1.3 x)."""
    from speechdrivestemplates_amd import ops
    from speechdrivestemplates_amd.core.networks import get_model
    K_STAGE = 2.0
    B, cfg_name = 32, "voice2pose_sdt_bp"
    ocfg = O.cfg_named(cfg_name)
    state = O.make_voice2pose_state(ocfg, N_CLIPS, seed=0, code_std=0.5)
    batch = O.make_batch(B, N_CLIPS, step=3, seed=11)
    st64 = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in state.items()}
    st32 = {k: v.detach() for k, v in state.items()}
    net = get_model("SequenceGeneratorCNN")(ocfg)
    net.load_state_dict({k[len("netG."):]: v.clone() for k, v in state.items() if k.startswith("netG.")}, strict=True)
    net.to(DEV).train()
    g = ocfg.VOICE2POSE.GENERATOR
    code32 = state["clips_code"][batch["clip_index"]].detach()
    rows = []  # (stage, hip_max, hip_rms, ref_max, ref_rms)

    def cl(x):  # logical channels-first -> channels-last on the device
        return ops.cl(x.to(DEV))

    def cf(x_cl):
        return ops.cf_view(x_cl).detach().cpu()

    with torch.no_grad():
        # ---- isolated stages on identical fp32 inputs (the float64 chain's activations, rounded)
        mel64 = O.mel_spectrogram(batch["audio"].double(), O.mel_window(torch.float64), O.mel_filterbank(torch.float64))
        from speechdrivestemplates_amd.mel import MelSpectrogram
        melT = MelSpectrogram().to(DEV)
        rows.append(("mel (unpinned oracle)",) + _stage_err(melT(batch["audio"].to(DEV)).cpu(), mel64) + _stage_err(O.mel_spectrogram(batch["audio"]), mel64))
        x64 = mel64.unsqueeze(1)
        for i, (_, _, _, s, p) in enumerate(O.AUDIO_ENCODER_2D):
            pre = "netG.audio_encoder.specgram_encoder_2d.%d.%d" % (i // 2, i % 2)
            xin = x64.float()
            ref64 = O.conv_norm_act(xin.double(), st64, pre, s, p, g.NORM, g.LEAKY_RELU, True)
            ref32 = O.conv_norm_act(xin, st32, pre, s, p, g.NORM, g.LEAKY_RELU, True)
            blk = net.audio_encoder.specgram_encoder_2d[i // 2][i % 2]
            hip = cf(blk.forward_cl(cl(xin) if i else xin.squeeze(1).unsqueeze(-1).to(DEV)))
            rows.append(("enc2d L%d alone" % i,) + _stage_err(hip, ref64) + _stage_err(ref32, ref64))
            x64 = ref64
        xin = x64.float()
        r64 = torch.cat([F.interpolate(xin.double(), (1, 64), mode="bilinear").squeeze(2), code32.double().unsqueeze(2).repeat(1, 1, 64)], 1)
        r32 = torch.cat([F.interpolate(xin, (1, 64), mode="bilinear").squeeze(2), code32.unsqueeze(2).repeat(1, 1, 64)], 1)
        hip = ops.ResizeConcatFn.apply(cl(xin), code32.to(DEV), 64)
        rows.append(("resize + code concat alone",) + _stage_err(cf(hip), r64) + _stage_err(r32, r64))
        xin = r64.float()
        u64 = O.unet_1d(st64, "netG.unet", xin.double(), g.NORM, g.LEAKY_RELU, True)
        u32 = O.unet_1d(st32, "netG.unet", xin, g.NORM, g.LEAKY_RELU, True)
        rows.append(("U-Net (12 Conv1d blocks) alone",) + _stage_err(cf(net.unet.forward_cl(cl(xin))), u64) + _stage_err(u32, u64))
        e64 = O._block1d(xin.double(), st64, "netG.unet.e0", False, g.NORM, g.LEAKY_RELU, True)
        e32 = O._block1d(xin, st32, "netG.unet.e0", False, g.NORM, g.LEAKY_RELU, True)
        rows.append(("  U-Net e0 (k3, Cin 288) alone",) + _stage_err(cf(net.unet.e0.forward_cl(cl(xin))), e64) + _stage_err(e32, e64))
        xin = u64.float()

        def dec(st, x):
            for j in range(4):
                x = O._block1d(x, st, "netG.decoder.%d" % j, False, g.NORM, g.LEAKY_RELU, True)
            return F.conv1d(x, st["netG.decoder.4.weight"], st["netG.decoder.4.bias"])

        from speechdrivestemplates_amd.core.networks.building_blocks import conv_head
        h = cl(xin)
        for blk in list(net.decoder)[:4]:
            h = blk.forward_cl(h)
        h = conv_head(h, net.decoder[4])
        d64 = dec(st64, xin.double())
        rows.append(("decoder (4 blocks + head) alone",) + _stage_err(cf(h), d64) + _stage_err(dec(st32, xin), d64))
        # ---- the cumulative chain, every implementation feeding itself from the same fp32 mel (the float64 mel, rounded)
        mel32 = mel64.float()
        p64 = O.generator(st64, "netG", mel32.double(), 64, code32.double(), ocfg, True)
        p32 = O.generator(st32, "netG", mel32, 64, code32, ocfg, True)
        ph = net(mel32.to(DEV), 64, code32.to(DEV)).cpu()
        x64c, x32c = mel32.double().unsqueeze(1), mel32.unsqueeze(1)
        xh = mel32.unsqueeze(-1).to(DEV)
        for i, (_, _, _, s, p) in enumerate(O.AUDIO_ENCODER_2D):
            pre = "netG.audio_encoder.specgram_encoder_2d.%d.%d" % (i // 2, i % 2)
            x64c = O.conv_norm_act(x64c, st64, pre, s, p, g.NORM, g.LEAKY_RELU, True)
            x32c = O.conv_norm_act(x32c, st32, pre, s, p, g.NORM, g.LEAKY_RELU, True)
            xh = net.audio_encoder.specgram_encoder_2d[i // 2][i % 2].forward_cl(xh)
            rows.append(("chain after enc2d L%d" % i,) + _stage_err(cf(xh), x64c) + _stage_err(x32c, x64c))
        rows.append(("chain: prediction",) + _stage_err(ph, p64) + _stage_err(p32, p64))
    torch.cuda.synchronize()
    lines = ["voice2pose_sdt_bp B=32 generator forward, per stage vs float64 (max-norm | rms), HIP and the fp32 oracle on identical inputs:"]
    bad = []
    for name, hm, hr, rm, rr in rows:
        lines.append("      %-34s hip %.3e | %.3e   ref32 %.3e | %.3e   ratio max %5.2f  rms %5.2f" % (name, hm, hr, rm, rr, hm / max(rm, 1e-30), hr / max(rr, 1e-30)))
        if "alone" in name and hr > (3.0 if name.startswith("mel") else K_STAGE) * rr + 2e-8:  # mel: a 512-long sum against an FFT
            bad.append((name, hr, rr))
        if name == "chain: prediction" and hm > K_FWD * rm + 2e-7:
            bad.append((name, hm, rm))
    _dump(lines)
    assert not bad, bad
