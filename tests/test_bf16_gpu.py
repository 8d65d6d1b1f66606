"""The bf16-storage path of BASELINE config 4 (voice2pose_sdt_bp, bf16): activations of the Conv2d chain and the conv operands' weight
copies live in HBM as bf16, products run on v_mfma_f32_32x32x16_bf16, accumulation / statistics / master weights / gradients are fp32.

The reference has no bf16 mode, so the tolerances are STATED here, per layer of the comparison:
  * one kernel against float64 on the SAME bf16-rounded operands: the products are exact in fp32, so what remains is the fp32 accumulation
    (1e-5 of max on the fp32 outputs: weight gradient, statistics) and ONE rounding of the output to bf16 (2^-8 = 3.9e-3 of each element's
    magnitude -> 5e-3 of max);
  * the bf16 chain / train step against the fp32 one and against the float64 oracle: prediction 4e-2 of max, losses 2e-2, every gradient
    tensor with cosine similarity >= 0.97 to the float64 gradient and within 40 % of its max-norm.  (The round-3 mode that only rounded the
    conv OPERANDS met 25 %; storing y, z, dz and dy of seven blocks as bf16 as well puts the two deepest tensors of the backward chain -- the
    first two encoder weights, 576 and 65536 elements -- at 0.32 / 0.26 of their max-norm with cosine 0.979: measured, stated, and what
    autocast-style bf16 training of this network costs.)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sdt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

LAYERS = [  # name, Hi, Wi, Cin, Cout, kh, kw, s, p : the seven Conv2d layers the bf16 kernels serve (generator.py:15-30)
    ("L1", 80, 427, 64, 64, 4, 4, 2, 1), ("L2", 40, 213, 64, 128, 3, 3, 1, 1), ("L3", 40, 213, 128, 128, 4, 4, 2, 1),
    ("L4", 20, 106, 128, 256, 3, 3, 1, 1), ("L5", 20, 106, 256, 256, 4, 4, 2, 1), ("L6", 10, 53, 256, 256, 3, 3, 1, 1),
    ("L7", 10, 53, 256, 256, 6, 3, 1, 0),
]


@pytest.fixture()
def ops():
    from speechdrivestemplates_amd import ops as _ops
    prev = _ops.STORAGE
    yield _ops
    _ops.set_storage(prev)
    assert not _ops.streamk_error_codes()


def relmax(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    assert torch.isfinite(a).all()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B", [3, 32])
@pytest.mark.parametrize("case", LAYERS, ids=[c[0] for c in LAYERS])
def test_bf16_conv_kernels_vs_float64_on_the_same_operands(ops, case, B):
    """sdt_convsk_bf16 forward (with the fused forward statistics) and input gradient, sdt_convsk_dw_bf16 (transposing LDS reads) against
    float64 convolutions of the SAME bf16-rounded tensors."""
    tag, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    if B == 32 and tag not in ("L1", "L4", "L7"):
        pytest.skip("full batch on three representative layers (float64 reference time)")
    g = torch.Generator().manual_seed(77 + Cin + kh + B)
    x = torch.randn(B, Hi, Wi, Cin, generator=g).to(BF)
    w = (torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(BF).float()  # bf16-representable master weights
    xd = x.to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w).to(DEV))
    groups = B
    assert ops.conv_stats_fusable(xd, wd, s, p, groups), "the bf16 stream-K plan must exist for every encoder layer"
    yd, sums = ops.ConvStatsFn.apply(xd, wd, s, p, groups, None)
    assert yd.dtype == BF
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    e = relmax(yd.float(), yr.permute(0, 2, 3, 1))
    assert e <= 5e-3, (tag, "forward", e)
    ref_sums = torch.stack([yr.sum((2, 3)), (yr * yr).sum((2, 3))], -1)  # (B, Cout, 2): taken from the fp32 accumulators, before rounding
    es = relmax(sums.view(B, Cout, 2), ref_sums)
    assert es <= 2e-5, (tag, "statistics", es)
    gy = torch.randn(yd.shape, generator=g).to(BF)
    gyd = gy.to(DEV)
    dxd = ops.conv_input_grad(gyd, wd, xd.shape, s, p)
    assert dxd.dtype == BF
    ops.conv_weight_grad(xd, gyd, wd, s, p)
    torch.cuda.synchronize()
    yr.backward(gy.double().permute(0, 3, 1, 2))
    edx = relmax(dxd.float(), xr.grad.permute(0, 2, 3, 1))
    edw = relmax(wd.grad, wr.grad)
    print("  %s B=%d bf16 kernels vs float64 on the same operands: fwd %.2e (stats %.1e)  dX %.2e  dW %.2e" % (tag, B, e, es, edx, edw))
    assert edx <= 5e-3, (tag, "dX", edx)
    assert edw <= 2e-5, (tag, "dW", edw)


@pytest.mark.parametrize("norm", ["IN", "BN"])
def test_bf16_input_gradient_with_backward_statistics(ops, norm):
    """EPI 2 on bf16 tensors: the input-gradient launch of a block also accumulates sum g, sum g*yhat of the normalisation below from the bf16
    forward output it reads back -- against float64 on the same rounded tensors."""
    from speechdrivestemplates_amd import _lib
    B, Hi, Wi, Cin, Cout, k, s, p = 3, 20, 53, 128, 256, 3, 1, 1
    g = torch.Generator().manual_seed(5)
    groups = B if norm == "IN" else 1
    ybelow = torch.randn(B, Hi, Wi, Cin, generator=g).to(BF)      # raw conv output of the block below
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(BF).float()
    gy = torch.randn(B, Hi, Wi, Cout, generator=g).to(BF)
    yb = ybelow.double()
    dims = (1, 2) if norm == "IN" else (0, 1, 2)
    mean = yb.mean(dims, keepdim=True)
    rstd = 1.0 / torch.sqrt(yb.var(dims, unbiased=False, keepdim=True) + 1e-5)
    h = ops.NormBwdHolder()
    h.y = ybelow.to(DEV)
    h.mean = mean.reshape(-1).float().to(DEV)
    h.rstd = rstd.reshape(-1).float().to(DEV)
    h.groups, h.slope = groups, 0.2
    wd = torch.nn.Parameter(ops.to_weight_layout(w).to(DEV))
    dx = ops.conv_input_grad(gy.to(DEV), wd, (B, Hi, Wi, Cin), s, p, h)
    torch.cuda.synchronize()
    assert h.sums is not None and dx.dtype == BF
    dxr = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.double(), None, s, p).permute(0, 2, 3, 1)
    assert relmax(dx.float(), dxr) <= 5e-3
    yh = (yb - mean) * rstd
    gg = dxr * torch.where(yh > 0, 1.0, 0.2)
    ref = torch.stack([gg.sum(dims), (gg * yh).sum(dims)], -1).reshape(groups, Cin, 2)
    e = relmax(h.sums.view(groups, Cin, 2), ref)
    print("  bf16 EPI 2 (%s): sums vs float64 %.2e" % (norm, e))
    assert e <= 1e-4, e
    assert _lib.BF16 == 1


def test_bf16_norm_and_first_block_io(ops):
    """sdt_colnorm_{fwd,bwd}_t and sdt_l0_block_{fwd,bwd}_t with bf16 tensors against the fp32 kernels on the same (rounded) inputs:
    identical arithmetic, the only difference is the final rounding of what is stored (2^-8 relative)."""
    g = torch.Generator().manual_seed(9)
    B, H, W, C = 3, 20, 53, 128
    y = torch.randn(B, H, W, C, generator=g).to(BF).to(DEV)
    gz = torch.randn(B, H, W, C, generator=g).to(BF).to(DEV)
    outs = {}
    for dt in (torch.float32, BF):
        yin = y.to(dt).requires_grad_(True)
        z = ops.ColNormActFn.apply(yin, None, None, None, None, None, B, 0.2, None, None)
        assert z.dtype == dt
        z.backward(gz.to(dt))
        torch.cuda.synchronize()
        outs[dt] = (z.detach().float(), yin.grad.detach().float())
    assert relmax(outs[BF][0], outs[torch.float32][0]) <= 5e-3
    assert relmax(outs[BF][1], outs[torch.float32][1]) <= 5e-3
    zf = ops.ColNormActFn.apply(y, None, None, None, None, None, B, 0.2, None, None, True)  # bf16 in, fp32 out (the block before the 1-D stage)
    assert zf.dtype == torch.float32 and relmax(zf, outs[torch.float32][0]) <= 1e-6
    # first block: fp32 mel -> bf16 z; backward reads a bf16 gradient
    mel = (torch.rand(B, 80, 427, generator=g) * 3.0).to(DEV)
    w0 = (torch.randn(64, 1, 3, 3, generator=g) * 0.3)
    gz0 = torch.randn(B, 80, 427, 64, generator=g).to(BF).to(DEV)
    res = {}
    for mode in ("f32", "bf16"):
        ops.set_storage(mode)
        wd = torch.nn.Parameter(ops.to_weight_layout(w0).to(DEV))
        z0 = ops.L0BlockFn.apply(mel, wd, None, None, None, None, None, B, 0.2, None)
        assert z0.dtype == (BF if mode == "bf16" else torch.float32)
        z0.backward(gz0.to(z0.dtype))
        torch.cuda.synchronize()
        res[mode] = (z0.detach().float(), wd.grad.detach().clone())
    ops.set_storage("f32")
    assert relmax(res["bf16"][0], res["f32"][0]) <= 5e-3
    assert relmax(res["bf16"][1], res["f32"][1]) <= 1e-5  # the same bf16 gradient values went in: the weight gradient is fp32 arithmetic either way


def test_bf16_encoder_chain_follows_the_fp32_chain(ops):
    """The audio encoder (8 blocks) in bf16 storage against the same weights in fp32 storage: output and every weight gradient."""
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.networks.keypoints_generation.generator import AudioEncoder
    from speechdrivestemplates_amd.optim import FlatAdam
    cfg = get_cfg_defaults()
    torch.manual_seed(3)
    enc = AudioEncoder(cfg).to(DEV).train()
    opt = FlatAdam(list(enc.parameters()))  # owns the weight mirrors and their bf16 copies
    g = torch.Generator().manual_seed(4)
    B = 4
    mel = (torch.rand(B, 80, 427, generator=g) ** 4 * 20.0).to(DEV)
    gout = torch.randn(B, 5, 51, 256, generator=g).to(DEV)
    res = {}
    for mode in ("f32", "bf16"):
        ops.set_storage(mode)
        opt.zero_grad()
        out = enc.encode_cl(mel)
        assert out.dtype == torch.float32
        out.backward(gout)
        ops.join_side_stream()
        torch.cuda.synchronize()
        res[mode] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in enc.named_parameters()})
    ops.set_storage("f32")
    e = relmax(res["bf16"][0], res["f32"][0])
    rows = []
    for k, ref in res["f32"][1].items():
        got = res["bf16"][1][k]
        cos = F.cosine_similarity(got.double().reshape(1, -1), ref.double().reshape(1, -1)).item()
        rows.append((k, relmax(got, ref), cos))
    print("  encoder bf16 vs fp32 storage: output %.2e; gradients worst rel %.2e, worst cosine %.5f"
          % (e, max(r[1] for r in rows), min(r[2] for r in rows)))
    assert e <= 4e-2, e
    assert max(r[1] for r in rows) <= 0.25 and min(r[2] for r in rows) >= 0.97, rows


@pytest.mark.parametrize("math1d", ["f32", "bf16"])
def test_b32_bf16_storage_vs_oracle(ops, math1d):
    """BASELINE config 4 at 32 clips per GPU: one forward + backward in bf16 storage against the float64 oracle (evaluated at the run's own L1
    sign decisions), at the stated bf16 bars: prediction 4e-2 of max, losses 2e-2, every gradient tensor within 40 % of its max-norm and
    with cosine similarity >= 0.97 to the float64 gradient (module docstring)."""
    from test_fullsize_gpu import N_CLIPS, _dump, _oracle_grads
    from test_model_gpu import _make_pipeline
    B, cfg_name = 32, "voice2pose_sdt_bp"
    ocfg = O.cfg_named(cfg_name)
    state = O.make_voice2pose_state(ocfg, N_CLIPS, seed=0, code_std=0.5)
    batch = O.make_batch(B, N_CLIPS, step=3, seed=11)
    ops.set_storage("bf16")
    ops.set_conv_math(math1d)  # 'bf16': the 1-D stage's conv products from bf16-rounded operands as well (fp32 tensors, fp32 accumulation)
    try:
        pipe, _ = _make_pipeline(cfg_name, N_CLIPS, 0.5)
        losses, results = pipe.forward_backward(batch)
        torch.cuda.synchronize()
    finally:
        ops.set_storage("f32")
        ops.set_conv_math("f32")
    grads_hip = {k: p.grad.detach().double().cpu() for k, p in pipe.model.named_parameters() if p.grad is not None}
    s_hip = torch.sign(results["poses_pred_normalized"].detach().cpu() - batch["poses"])
    l64, p64, g64 = _oracle_grads(cfg_name, state, batch, torch.float64, l1_signs=s_hip)
    e = relmax(results["poses_pred_normalized"], p64)
    rows = []
    for k, ref in g64.items():
        got = grads_hip[k]
        rel = ((got - ref).abs().max() / ref.abs().max()).item()
        cos = (F.cosine_similarity(got.reshape(1, -1), ref.reshape(1, -1)).item()) if ref.numel() > 1 else 1.0
        rows.append((k, rel, cos))
    lines = ["voice2pose_sdt_bp B=32 (bf16 STORAGE, 1-D stage products %s) vs float64 oracle: prediction rel-max-err %.3e; losses %s" % (
        math1d, e, {k: "%.2e" % abs(float(losses[k].detach()) / float(l64[k]) - 1.0) for k in ("G_reg_loss", "G_clipcode_kl_loss", "G_loss")})]
    lines += ["      %-60s rel-max-err %.3e  cosine %.5f" % r for r in rows]
    _dump(lines)
    assert e <= 4e-2, e
    for k in ("G_reg_loss", "G_clipcode_kl_loss", "G_loss"):
        a, b = float(losses[k]), float(l64[k])
        assert abs(a - b) <= 2e-2 * abs(b), (k, a, b)
    worst_rel, worst_cos = max(r[1] for r in rows), min(r[2] for r in rows)
    print("  bf16 storage B=32: worst gradient rel-max-err %.3e, worst cosine %.5f" % (worst_rel, worst_cos))
    assert worst_rel <= 0.40 and worst_cos >= 0.97, [r for r in rows if r[1] > 0.40 or r[2] < 0.97]


def test_bf16_storage_train_steps_track_the_fp32_run(ops):
    """Three train steps (forward, backward, Adam) in bf16 storage stay within 2 % of the fp32 run's losses, repeat bit-identically, and leave
    no stream-K error word."""
    from test_model_gpu import _make_pipeline
    hist = {}
    for mode in ("f32", "bf16", "bf16"):
        ops.set_storage(mode)
        try:
            pipe, _ = _make_pipeline("voice2pose_sdt_bp", 64, 0.5)
            h = []
            for step in range(3):
                batch = O.make_batch(8, 64, step=step, seed=1)
                losses, _ = pipe.forward_backward(batch)
                pipe.optimizer_updates(losses)
                torch.cuda.synchronize()
                h.append((float(losses["G_loss"]), float(losses["G_reg_loss"])))
            w = pipe.model.netG.audio_encoder.specgram_encoder_2d[2][0].conv.weight.detach().clone()
        finally:
            ops.set_storage("f32")
        hist.setdefault(mode, []).append((h, w))
    (hf, _), = hist["f32"]
    (h1, w1), (h2, w2) = hist["bf16"]
    for (a, b), (c, d) in zip(hf, h1):
        assert abs(a - c) <= 2e-2 * abs(a) and abs(b - d) <= 2e-2 * abs(b), (hf, h1)
    assert h1 == h2 and torch.equal(w1, w2), "the bf16-storage path is not run-to-run deterministic"


def test_bf16_storage_blocks_round_where_the_emulating_oracle_rounds(ops):
    """VERDICT r4 item 6c.  The bars of test_b32_bf16_storage_vs_oracle (prediction 4e-2, gradients 40 %) are set by bf16 rounding itself: the float64
    oracle they compare with rounds nothing.  oracle.sdt_oracle.BF16_EMULATION rounds EXACTLY where the bf16-storage path rounds (block 0: its output;
    blocks 1-7: the weights, the conv output as stored -- statistics from the unrounded accumulators --, the activated output; the last block writes
    fp32).  Rounding is a discontinuity, so two computations that agree to fp32 precision decorrelate within three or four layers (a difference eps in
    a value flips its rounding with probability eps / ulp: 4e-5 -> 2e-4 -> 7e-4 -> 2e-3 along the encoder, tools/debug/bf16_emu_model.py) -- an
    end-to-end comparison cannot be tightened this way.  What CAN be checked tightly is every block on ITS OWN input: the engine's eight encoder blocks
    run in bf16 storage on the real mel and weights of the B = 8 batch, and each block's output is compared with the emulating oracle block applied to
    the engine's input of that block.  Bars (4 x measured): relative RMS difference <= 1.5e-4 (measured 1.6e-5 .. 3.5e-5: 4e-5 .. 9e-5 of the stored values
    land on the other side of a rounding boundary, by one ulp) AND at most a twentieth of the distance to the oracle that does not round (1.7e-3 .. 3.1e-3)."""
    from test_model_gpu import _make_pipeline
    B, N, cfg_name = 8, 64, "voice2pose_sdt_bp"
    ocfg = O.cfg_named(cfg_name)
    state = O.make_voice2pose_state(ocfg, N, seed=0, code_std=0.5)
    batch = O.make_batch(B, N, step=3, seed=11)
    st64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in state.items()}

    def rms(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

    ops.set_storage("bf16")
    try:
        pipe, _ = _make_pipeline(cfg_name, N, 0.5)
        net = pipe.model.netG
        with torch.no_grad():
            x = pipe.model.mel_transfm(batch["audio"].to(DEV)).unsqueeze(-1)  # (B, H, W, 1) channels-last
            rows, i = [], 0
            for stage in net.audio_encoder.specgram_encoder_2d:
                for block in stage:
                    y = block.forward_cl(x, None, None, out_f32=(i == 7))
                    assert y.dtype == (torch.float32 if i == 7 else torch.bfloat16), (i, y.dtype)
                    xin = x.float().double().cpu().permute(0, 3, 1, 2)  # exactly the values the engine's block read
                    s_, p_ = O.AUDIO_ENCODER_2D[i][3], O.AUDIO_ENCODER_2D[i][4]
                    prefix = "netG.audio_encoder.specgram_encoder_2d.%d.%d" % (i // 2, i % 2)
                    emu = "l0" if i == 0 else ("2d_last" if i == 7 else "2d")
                    o_emu = O.conv_norm_act(xin, st64, prefix, s_, p_, "IN", True, True, emu)
                    o_plain = O.conv_norm_act(xin, st64, prefix, s_, p_, "IN", True, True, None)
                    got = y.float().permute(0, 3, 1, 2)
                    differing = (got.double().cpu() != o_emu).double().mean().item()
                    rows.append((i, rms(got, o_emu), rms(got, o_plain), differing))
                    x, i = y, i + 1
    finally:
        ops.set_storage("f32")
    for i, e_emu, e_plain, frac in rows:
        print("  encoder block %d in bf16 storage on its own input: rms difference to the emulating oracle %.2e (fraction of differing values %.1e), "
              "to the oracle that does not round %.2e" % (i, e_emu, frac, e_plain))
    for i, e_emu, e_plain, frac in rows:
        if i == 7:  # fp32 output: nothing is rounded after the normalisation; what is left is the stored conv output's rounding flips
            assert e_emu <= 1.5e-4, (i, e_emu)
            continue
        assert e_emu <= 1.5e-4 and e_emu <= 0.05 * e_plain, (i, e_emu, e_plain)
        assert frac <= 5e-4, (i, frac)


def test_bf16_storage_block_backward_rounds_where_the_emulating_oracle_rounds(ops):
    """the backward-side counterpart of the test above.  The step-level bf16 bars (prediction 4e-2, every gradient within 40 % of
    max-norm, cosine >= 0.97: test_b32_bf16_storage_vs_oracle) are set by bf16 rounding itself and cannot be tightened end to end (rounding decorrelates
    two fp32-equivalent computations within three or four layers).  Per BLOCK on the engine's own tensors they can: every encoder block 1-7 runs forward
    in bf16 storage on the engine's input of that block, then backward on a bf16 gradient of its output; the three things the backward produces are
    compared with a float64 emulation that rounds exactly where the kernels round --
        dy  = IN-backward of g = gz * act'(yhat), yhat from the STORED (bf16) conv output and the statistics of the unrounded one   -> stored bf16
        dX  = conv_transpose(dy_bf16, W_bf16), fp32 accumulation                                                                     -> stored bf16
        dW  = conv weight gradient of (x_bf16, dy_bf16), fp32 accumulation                                                           -> fp32
    -- and with the same float64 formulas WITHOUT the two roundings (the distance bf16 storage itself costs).  Bars: relative RMS difference to the
    emulation <= 3e-4 for dX and dW (what is left are one-ulp rounding flips of a few 1e-4 of the stored dy / dX values) AND at most a fifth of the
    distance to the formulas that do not round."""
    from test_model_gpu import _make_pipeline
    B, N, cfg_name = 4, 64, "voice2pose_sdt_bp"
    batch = O.make_batch(B, N, step=3, seed=11)

    def rms(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

    def q(t):  # round to bf16, keep float64
        return t.float().to(torch.bfloat16).double()

    rows = []
    ops.set_storage("bf16")
    try:
        pipe, _ = _make_pipeline(cfg_name, N, 0.5)
        net = pipe.model.netG
        blocks = [b for stage in net.audio_encoder.specgram_encoder_2d for b in stage]
        with torch.no_grad():
            x = pipe.model.mel_transfm(batch["audio"].to(DEV)).unsqueeze(-1)
            x = blocks[0].forward_cl(x, None, None)
        gen = torch.Generator().manual_seed(77)
        for i in range(1, 8):
            blk = blocks[i]
            pipe.optimizers["optimizerG"].zero_grad()
            xin = x.detach().clone().requires_grad_(True)
            holder = ops.NormBwdHolder()
            y = blk.forward_cl(xin, None, holder, out_f32=(i == 7))
            gz = (torch.randn(y.shape, generator=gen) * 1e-3).to(DEV).to(y.dtype)
            y.backward(gz)
            torch.cuda.synchronize()
            dx_eng = xin.grad.detach().float().cpu()
            # ---- the emulation, float64 on the CPU, from the ENGINE's stored tensors
            ys = holder.y.detach().double().cpu()                      # (B, Ho, Wo, C) as stored (bf16)
            Bq, Ho, Wo, C = ys.shape
            mean = holder.mean.detach().double().cpu().reshape(Bq, 1, 1, C)
            rstd = holder.rstd.detach().double().cpu().reshape(Bq, 1, 1, C)
            yhat = (ys - mean) * rstd
            g = gz.detach().double().cpu() * torch.where(yhat > 0, torch.ones_like(yhat), torch.full_like(yhat, blk.slope))
            m1, m2 = g.mean(dim=(1, 2), keepdim=True), (g * yhat).mean(dim=(1, 2), keepdim=True)
            dy = rstd * (g - m1 - yhat * m2)
            w_log = blk.conv.weight.detach()                           # logical (Cout, Cin, kh, kw), fp32 master
            res = {}
            for name, rnd in (("emu", True), ("plain", False)):
                xa = xin.detach().double().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)   # the bf16 input the engine read
                wa = (q(w_log.cpu()) if rnd else w_log.double().cpu()).clone().requires_grad_(True)
                ya = torch.nn.functional.conv2d(xa, wa, None, blk.stride, blk.padding)
                ya.backward((q(dy) if rnd else dy).permute(0, 3, 1, 2))
                res[name] = ((q(xa.grad) if rnd else xa.grad).permute(0, 2, 3, 1), wa.grad)
            dw_eng = blk.conv.weight.grad.detach().double().cpu()      # logical layout view of the flat gradient buffer
            rows.append((i, rms(dx_eng, res["emu"][0]), rms(dx_eng, res["plain"][0]), rms(dw_eng, res["emu"][1]), rms(dw_eng, res["plain"][1])))
            with torch.no_grad():
                x = y.detach()
    finally:
        ops.set_storage("f32")
    for i, ex, px, ew, pw in rows:
        print("  encoder block %d backward in bf16 storage: dX rms difference to the emulation %.2e (to the formulas that do not round %.2e), "
              "dW %.2e (%.2e)" % (i, ex, px, ew, pw))
    for i, ex, px, ew, pw in rows:
        assert ex <= 3e-4 and ex <= 0.2 * px, (i, "dX", ex, px)
        assert ew <= 3e-4 and ew <= 0.2 * pw, (i, "dW", ew, pw)
