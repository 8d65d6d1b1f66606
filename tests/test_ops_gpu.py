"""GPU parity tests, one per kernel family: each libsdt_hip.so entry point (called through the C ABI via
speechdrivestemplates_amd.ops) against a float64 torch-CPU statement of the same reference operator on
seeded inputs.  Tolerances are relative to the reference's max magnitude and written next to each check."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import REPO

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), "non-finite values in kernel output"
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def check(name, got, ref, tol):
    """`tol`: the stated fp32 tolerance of this quantity; the check is ALSO held to 10x the error recorded in tests/golden/margins.json
    (conftest.calibrated_bound), so that a 10x numerical regression fails even where the stated tolerance is generous."""
    from conftest import calibrated_bound
    e = rel_err(got, ref)
    bound = calibrated_bound(name, e, tol)
    print("  %-40s rel-max-err %.3e (tol %.1e, held to %.1e)" % (name, e, tol, bound))
    assert e < bound, "%s: %.3e >= %.1e (stated tolerance %.1e)" % (name, e, bound, tol)


@pytest.fixture(scope="module")
def ops():
    from speechdrivestemplates_amd import ops as o
    return o


# every distinct conv shape on the hot path: (tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p) -- generator.py:15-30
CONV2D = [
    ("L0 1->64 k3", 2, 80, 427, 1, 64, 3, 3, 1, 1),
    ("L1 64->64 k4s2", 2, 80, 427, 64, 64, 4, 4, 2, 1),
    ("L2 64->128 k3", 2, 40, 213, 64, 128, 3, 3, 1, 1),
    ("L3 128->128 k4s2", 2, 40, 213, 128, 128, 4, 4, 2, 1),
    ("L4 128->256 k3", 2, 20, 106, 128, 256, 3, 3, 1, 1),
    ("L5 256->256 k4s2", 2, 20, 106, 256, 256, 4, 4, 2, 1),
    ("L6 256->256 k3", 2, 10, 53, 256, 256, 3, 3, 1, 1),
    ("L7 256->256 k(6,3)p0", 2, 10, 53, 256, 256, 6, 3, 1, 0),
    ("big-M 128x128 tile", 8, 40, 213, 64, 128, 3, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV2D, ids=[c[0] for c in CONV2D])
def test_conv2d(ops, case):
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) * (2.0 / (Cin * kh * kw)) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd = ops.cl(x.float()).to(DEV).requires_grad_(True)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    assert ops.weight_storage(wd).data_ptr() == wd.data_ptr()
    yd = ops.ConvFn.apply(xd, wd, None, s, p)
    yd.backward(ops.cl(gy.float()).to(DEV))
    torch.cuda.synchronize()
    check(tag + " fwd", ops.cf_view(yd), y, 2e-5)
    check(tag + " dX", ops.cf_view(xd.grad), xr.grad, 2e-5)
    check(tag + " dW", wd.grad, wr.grad, 5e-5)
    # a second backward accumulates into .grad (standard autograd semantics)
    yd2 = ops.ConvFn.apply(xd.detach(), wd, None, s, p)
    yd2.backward(ops.cl(gy.float()).to(DEV))
    check(tag + " dW accumulate", wd.grad, 2 * wr.grad, 5e-5)


@pytest.mark.parametrize("mode,tol", [("f32", 3e-6), ("bf16x6", 3e-6), ("bf16x3", 1e-4), ("bf16", 1.5e-2)])
def test_conv_math_modes(ops, mode, tol):
    """Opt-in product arithmetic of the forward / input-gradient kernels (sdt_set_conv_math): fp32 in and out in every
    mode; 'bf16x6' (exact 3-piece split of both operands, 6 bf16 MFMA products) must be as accurate as the exact-fp32
    MFMA kernel; 'bf16' is BASELINE config 4's precision.  Error is max|d| / max|ref| against float64."""
    g = torch.Generator().manual_seed(77)
    cases = [(2, 20, 53, 128, 256, 3, 3, 1, 1), (2, 21, 40, 64, 64, 4, 4, 2, 1), (3, 1, 64, 256, 256, 1, 3, 1, 1)]
    prev = ops.set_conv_math(mode)
    try:
        assert prev == "f32"
        for B, Hi, Wi, Cin, Cout, kh, kw, s, p in cases:
            x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
            w = torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) * (2.0 / (Cin * kh * kw)) ** 0.5
            xr = x.clone().requires_grad_(True)
            y = F.conv2d(xr, w, None, s, p)
            gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
            y.backward(gy)
            wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
            xd = ops.cl(x.float()).to(DEV)
            yd = ops.conv_forward(xd, wd, None, s, p)
            dxd = ops.conv_input_grad(ops.cl(gy.float()).to(DEV), wd, xd.shape, s, p)
            check("%s fwd" % mode, ops.cf_view(yd), y, tol)
            check("%s dX" % mode, ops.cf_view(dxd), xr.grad, tol)
    finally:
        ops.set_conv_math("f32")
    with pytest.raises(KeyError):
        ops.set_conv_math("fp8")


CONV1D = [  # (tag, B, T, Cin, Cout, k, s, p, bias)
    ("e0 288->256 k3", 4, 64, 288, 256, 3, 1, 1, False),
    ("e2 256->256 k4s2", 4, 64, 256, 256, 4, 2, 1, False),
    ("e6 T4->2", 4, 4, 256, 256, 4, 2, 1, False),
    ("d5 T4 k3", 4, 4, 256, 256, 3, 1, 1, False),
    ("head 256->242 k1 bias", 4, 64, 256, 242, 1, 1, 0, True),
    ("pose-enc 242->256 k3", 4, 64, 242, 256, 3, 1, 1, False),
    ("pose-enc 256->64 k4s2", 4, 4, 256, 64, 4, 2, 1, False),
    ("disc 242->256 k4s2 T63", 4, 63, 242, 256, 4, 2, 1, False),
    ("disc 512->1024 k3 T15", 4, 15, 512, 1024, 3, 1, 1, False),
    ("disc head 1024->1 k3 bias", 4, 15, 1024, 1, 3, 1, 1, True),
    ("ae d5 32->256 k3 T4", 4, 4, 32, 256, 3, 1, 1, False),
    ("T odd 37 k4s2", 3, 37, 256, 256, 4, 2, 1, False),
]


@pytest.mark.parametrize("case", CONV1D, ids=[c[0] for c in CONV1D])
def test_conv1d(ops, case):
    tag, B, T, Cin, Cout, k, s, p, has_bias = case
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    x = torch.randn(B, Cin, T, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, generator=g, dtype=torch.float64) * (2.0 / (Cin * k)) ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64) if has_bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_bias else None
    y = F.conv1d(xr, wr, br, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd = ops.cl(x.float()).to(DEV).requires_grad_(True)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    bd = torch.nn.Parameter(b.float().to(DEV)) if has_bias else None
    yd = ops.ConvFn.apply(xd, wd, bd, s, p)
    yd.backward(ops.cl(gy.float()).to(DEV))
    torch.cuda.synchronize()
    check(tag + " fwd", ops.cf_view(yd), y, 2e-5)
    check(tag + " dX", ops.cf_view(xd.grad), xr.grad, 2e-5)
    check(tag + " dW", wd.grad, wr.grad, 5e-5)
    if has_bias:
        check(tag + " dbias", bd.grad, br.grad, 2e-5)


@pytest.mark.parametrize("shape", [(2, 80, 427, 64), (2, 5, 51, 256), (3, 20, 106, 128)])
def test_instance_norm2d_leaky(ops, shape):
    B, H, W, C = shape
    g = torch.Generator().manual_seed(C + H)
    y = (torch.randn(B, C, H, W, generator=g, dtype=torch.float64) * 3 + 1.5).requires_grad_(True)
    z = F.leaky_relu(F.instance_norm(y, eps=1e-5), 0.2)
    gz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(gz)
    yd = ops.cl(y.detach().float()).to(DEV).requires_grad_(True)
    zd = ops.ColNormActFn.apply(yd, None, None, None, None, None, B, 0.2)
    zd.backward(ops.cl(gz.float()).to(DEV))
    check("IN2d fwd %s" % (shape,), ops.cf_view(zd), z, 1e-5)
    check("IN2d bwd %s" % (shape,), ops.cf_view(yd.grad), y.grad, 2e-5)


@pytest.mark.parametrize("norm", ["IN", "BN"])
@pytest.mark.parametrize("hw", [(80, 427), (80, 300), (7, 5), (81, 37, 33)])
def test_fused_first_block(ops, norm, hw):
    """Conv2d(1,64,k3,p1) + IN2d|BN2d(train) + LeakyReLU fused (stats from mel moments, yhat recomputed in backward).
    (81, 37) at B = 33: the backward kernel takes TWO image rows per workgroup there (as at the bench size), the last workgroup of a clip
    has one, and with 3 columns per thread segment the last segments lie partly or wholly past the row (masked pixels, clamped loads)."""
    H, W = hw[:2]
    B = hw[2] if len(hw) > 2 else 3
    g = torch.Generator().manual_seed(H * W)
    mel = (torch.rand(B, H, W, generator=g, dtype=torch.float64) ** 3) * 40.0  # power-mel like: non-negative, heavy tail
    w = (torch.randn(64, 1, 3, 3, generator=g, dtype=torch.float64) * (2.0 / 9) ** 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(64, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(64, generator=g, dtype=torch.float64)).requires_grad_(True)
    rm, rv = torch.zeros(64, dtype=torch.float64), torch.ones(64, dtype=torch.float64)
    y = F.conv2d(mel.unsqueeze(1), w, None, 1, 1)
    u = F.instance_norm(y, eps=1e-5) if norm == "IN" else F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    z = F.leaky_relu(u, 0.2)
    gz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(gz)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.detach().float()).to(DEV))
    if norm == "IN":
        zd = ops.L0BlockFn.apply(mel.float().to(DEV), wd, None, None, None, None, None, B, 0.2)
    else:
        gd, bd = torch.nn.Parameter(gamma.detach().float().to(DEV)), torch.nn.Parameter(beta.detach().float().to(DEV))
        rmd, rvd = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        zd = ops.L0BlockFn.apply(mel.float().to(DEV), wd, gd, bd, rmd, rvd, nbt, 1, 0.2)
    zd.backward(ops.cl(gz.float()).to(DEV))
    tag = "L0 fused %s %s" % (norm, hw)
    # gradient sums run over up to 6.5 M elements: an element whose pre-activation is within fp32 rounding of the
    # LeakyReLU kink may take the other slope than in the float64 reference (a 0.8*|dz| change of one term)
    check(tag + " fwd", ops.cf_view(zd), z, 2e-5)
    check(tag + " dW", wd.grad, w.grad, 2e-3)
    if norm == "BN":
        check(tag + " running_mean", rmd, rm, 1e-5)
        check(tag + " running_var", rvd, rv, 1e-5)
        assert int(nbt.item()) == 1
        check(tag + " dgamma", gd.grad, gamma.grad, 2e-3)
        check(tag + " dbeta", bd.grad, beta.grad, 2e-3)


@pytest.mark.parametrize("slope", [0.2, 0.0])
@pytest.mark.parametrize("shape", [(4, 64, 256), (4, 2, 64), (2, 40 * 213, 64), (4, 15, 1024)])
def test_batch_norm_train_act(ops, shape, slope):
    B, T, C = shape
    g = torch.Generator().manual_seed(C + T)
    y = (torch.randn(B, C, T, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    u = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    z = F.leaky_relu(u, slope) if slope else F.relu(u)
    gz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(gz)
    yd = ops.cl(y.detach().float()).to(DEV).requires_grad_(True)
    gd = torch.nn.Parameter(gamma.detach().float().to(DEV))
    bd = torch.nn.Parameter(beta.detach().float().to(DEV))
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    zd = ops.ColNormActFn.apply(yd, gd, bd, rmd, rvd, nbt, 1, slope)
    zd.backward(ops.cl(gz.float()).to(DEV))
    tag = "BN %s slope %.1f" % (shape, slope)
    check(tag + " fwd", ops.cf_view(zd), z, 1e-5)
    check(tag + " running_mean", rmd, rm, 1e-5)
    check(tag + " running_var", rvd, rv, 1e-5)
    assert int(nbt.item()) == 1
    check(tag + " dy", ops.cf_view(yd.grad), y.grad, 5e-5)
    check(tag + " dgamma", gd.grad, gamma.grad, 5e-5)
    check(tag + " dbeta", bd.grad, beta.grad, 5e-5)
    # eval mode uses the running statistics
    ze = F.leaky_relu(F.batch_norm(y.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 1e-5), slope)
    zde = ops.colnorm_eval(yd.detach(), gd.detach(), bd.detach(), rmd, rvd, slope)
    check(tag + " eval", ops.cf_view(zde), ze, 1e-5)


@pytest.mark.parametrize("shape", [(4, 64, 256), (4, 2, 256), (3, 7, 512), (2, 5, 64)])
def test_rownorm_leaky(ops, shape):
    B, T, C = shape
    g = torch.Generator().manual_seed(C * T)
    y = (torch.randn(B, C, T, generator=g, dtype=torch.float64) * 2 + 0.3).requires_grad_(True)
    # the reference's InstanceNorm1d on the permuted tensor (building_blocks.py:50-51)
    z = F.leaky_relu(F.instance_norm(y.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1), 0.2)
    gz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(gz)
    yd = ops.cl(y.detach().float()).to(DEV).requires_grad_(True)
    zd = ops.RowNormActFn.apply(yd, 0.2)
    zd.backward(ops.cl(gz.float()).to(DEV))
    check("rownorm fwd %s" % (shape,), ops.cf_view(zd), z, 1e-5)
    check("rownorm bwd %s" % (shape,), ops.cf_view(yd.grad), y.grad, 2e-5)


@pytest.mark.parametrize("T,D", [(64, 32), (64, 0), (40, 32), (360, 32)])
def test_resize_concat(ops, T, D):
    B, H, W, C = 3, 5, 51 if T != 360 else 283, 256
    g = torch.Generator().manual_seed(T + D)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    code = torch.randn(B, D, generator=g, dtype=torch.float64, requires_grad=True) if D else None
    r = F.interpolate(x, (1, T), mode="bilinear").squeeze(2)
    if D:
        r = torch.cat([r, code.unsqueeze(2).repeat(1, 1, T)], 1)
    gr = torch.randn(r.shape, generator=g, dtype=torch.float64)
    r.backward(gr)
    xd = ops.cl(x.detach().float()).to(DEV).requires_grad_(True)
    cd = code.detach().float().to(DEV).requires_grad_(True) if D else None
    rd = ops.ResizeConcatFn.apply(xd, cd, T)
    rd.backward(ops.cl(gr.float()).to(DEV))
    # source indices / lambdas are computed in fp32 by the reference too (ATen area_pixel_compute_source_index):
    # tight against the fp32 reference, looser against fp64 when W/T is not exactly representable
    r32 = F.interpolate(x.detach().float(), (1, T), mode="bilinear").squeeze(2)
    check("resize fwd vs fp32 ref T%d" % T, ops.cf_view(rd)[:, :C], r32, 2e-6)
    check("resize+concat fwd T%d D%d" % (T, D), ops.cf_view(rd), r, 5e-5)
    check("resize bwd dx", ops.cf_view(xd.grad), x.grad, 5e-5)
    if D:
        check("resize bwd dcode", cd.grad, code.grad, 1e-5)


@pytest.mark.parametrize("Ti,To,skip", [(2, 4, True), (32, 64, True), (4, 8, False), (45, 90, True), (10, 23, True)])
def test_upsample_add(ops, Ti, To, skip):
    B, C = 3, 256
    g = torch.Generator().manual_seed(Ti + To)
    prev = torch.randn(B, C, Ti, generator=g, dtype=torch.float64, requires_grad=True)
    sk = torch.randn(B, C, To, generator=g, dtype=torch.float64, requires_grad=True) if skip else None
    r = F.interpolate(prev, To, mode="linear")
    if skip:
        r = r + sk
    gr = torch.randn(r.shape, generator=g, dtype=torch.float64)
    r.backward(gr)
    pd = ops.cl(prev.detach().float()).to(DEV).requires_grad_(True)
    sd = ops.cl(sk.detach().float()).to(DEV).requires_grad_(True) if skip else None
    rd = ops.UpsampleAddFn.apply(pd, sd, To)
    rd.backward(ops.cl(gr.float()).to(DEV))
    check("upsample+add fwd %d->%d" % (Ti, To), ops.cf_view(rd), r, 5e-6)
    check("upsample bwd dprev", ops.cf_view(pd.grad), prev.grad, 5e-5)
    if skip:
        check("upsample bwd dskip", ops.cf_view(sd.grad), sk.grad, 1e-6)


def test_l1_loss(ops):
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(4, 64, 2, 121, generator=g, dtype=torch.float64, requires_grad=True)
    gt = torch.randn(4, 64, 2, 121, generator=g, dtype=torch.float64)
    gt[0, 0, 0, :5] = pred.detach()[0, 0, 0, :5]  # exact ties: sign(0) = 0
    loss = (torch.abs(pred - gt) * 1.0).mean()
    (loss * 1.7).backward()
    pd = pred.detach().float().to(DEV).requires_grad_(True)
    ld = ops.L1LossFn.apply(pd, gt.float().to(DEV), 1.0)
    (ld * 1.7).backward()
    check("L1 loss", ld, loss, 1e-6)
    check("L1 grad", pd.grad, pred.grad, 1e-6)


def test_code_gather_kl(ops):
    from oracle import sdt_oracle as O
    g = torch.Generator().manual_seed(6)
    N, B, D = 50, 8, 32
    table = torch.randn(N, D, generator=g, dtype=torch.float64) * 0.7
    idx = torch.tensor([3, 17, 4, 49, 0, 21, 8, 30])
    tr = table.clone().requires_grad_(True)
    code = tr[idx]
    kl = O.clip_code_kl(code, 0.1)
    gcode = torch.randn(B, D, generator=g, dtype=torch.float64)
    (kl * 1.3 + (code * gcode).sum()).backward()
    td = torch.nn.Parameter(table.float().to(DEV))
    cd, kd, valid = ops.CodeGatherKLFn.apply(td, idx.to(DEV), 0.1)
    (kd * 1.3 + (cd * gcode.float().to(DEV)).sum()).backward()
    assert int(valid.item()) == 1
    check("code gather", cd, code, 1e-7)
    check("code KL", kd, kl, 1e-5)
    check("code table grad (dense)", td.grad, tr.grad, 1e-5)
    # zero-variance column -> term skipped: loss 0, no KL gradient (voice2pose.py:154)
    t0 = torch.zeros(N, D)
    td0 = torch.nn.Parameter(t0.to(DEV))
    c0, k0, v0 = ops.CodeGatherKLFn.apply(td0, idx.to(DEV), 0.1)
    (k0 + (c0 * gcode.float().to(DEV)).sum()).backward()
    assert int(v0.item()) == 0 and float(k0.item()) == 0.0
    ref = torch.zeros(N, D, dtype=torch.float64)
    ref[idx] = gcode
    check("zero-var: only gather grad", td0.grad, ref, 1e-7)
    assert O.clip_code_kl(t0[idx], 0.1) is None
    # a clip index beyond the table (an EXTERNAL_CODE table from a smaller checkpoint): the reference raises IndexError; the
    # kernels must not touch memory outside the table -- the gathered row is NaN (loud) and its gradient row is dropped
    bad = idx.clone()
    bad[2] = N + 7
    td1 = torch.nn.Parameter(table.float().to(DEV))
    guard = torch.full((4 * N * D,), 3.0, device=DEV)  # the table's neighbourhood in the allocator: must stay untouched
    c1, k1, v1 = ops.CodeGatherKLFn.apply(td1, bad.to(DEV), 0.1)
    assert torch.isnan(c1[2]).all() and torch.isfinite(c1[[0, 1, 3, 4, 5, 6, 7]]).all() and torch.isnan(k1)
    (c1 * gcode.float().to(DEV)).sum().backward()
    ok = [i for i in range(B) if i != 2]
    ref1 = torch.zeros(N, D, dtype=torch.float64)
    ref1[idx[ok]] = gcode[ok]
    check("out-of-range row: gradient dropped", td1.grad, ref1, 1e-7)
    assert bool((guard == 3.0).all())
    # B == 1 outside training (last validation batch of one clip): the reference's torch.var gives nan and the nan KL term is
    # added (voice2pose.py:152-157) -- no exception; in training it stays a configuration error
    c2, k2, v2 = ops.CodeGatherKLFn.apply(table.float().to(DEV), idx[:1].to(DEV), 0.1, True)
    assert torch.equal(c2.cpu(), table.float()[idx[:1]]) and torch.isnan(k2) and int(v2) == 1
    with pytest.raises(RuntimeError):
        ops.CodeGatherKLFn.apply(td, idx[:1].to(DEV), 0.1)


def test_final_metrics(ops):
    from oracle import sdt_oracle as O
    batch = O.make_batch(4, 16, step=2, seed=3)
    g = torch.Generator().manual_seed(7)
    pred = torch.randn(4, 64, 2, 121, generator=g)
    st = batch["speaker_stat"]
    for hier in (True, False):
        fp = O.get_final_results(pred.clone(), st, hier)
        fg = O.get_final_results(batch["poses"].clone(), st, hier)
        m = O.evaluate_step(fp, fg)
        dfp, dfg, dm = ops.final_metrics(pred.to(DEV), batch["poses"].to(DEV), st["mean"].to(DEV), st["std"].to(DEV),
                                         st["scale_factor"].to(DEV), hier)
        assert dfp.dtype == torch.float64
        check("final pred hier=%s" % hier, dfp, fp, 1e-12)
        check("final gt", dfg, fg, 1e-12)
        check("L2_dist", dm[0], m["L2_dist"], 1e-10)
        check("lip_sync_error_n", dm[1], m["lip_sync_error_n"], 1e-10)


def test_adam_matches_torch(ops):
    g = torch.Generator().manual_seed(8)
    n = 100003
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().double().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    pd = torch.zeros(((n + 3) // 4) * 4, device=DEV)
    pd[:n] = p0.to(DEV)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    lr = torch.tensor([1e-3], device=DEV)
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    for step in range(4):
        gr = torch.randn(n, generator=g) * (0.0 if step == 2 else 1.0)
        pr.grad = gr.double()
        opt.step()
        gd = torch.zeros_like(pd)
        gd[:n] = gr.to(DEV)
        ops.adam_step(pd[:n], gd[:n], m[:n], v[:n], lr, state)
    assert int(state[0].item()) == 4
    check("Adam params after 4 steps", pd[:n], pr, 1e-6)


def test_weight_mirrors_stay_current(ops):
    """The batched (Cin,taps,Cout) weight mirrors behind conv_input_grad: refreshed after an optimiser step, after a
    torch-level in-place edit (version counter), and dropped with their optimiser."""
    from speechdrivestemplates_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(21)
    ws = [torch.nn.Parameter(ops.to_weight_layout((torch.randn(shape, generator=g) * 0.1).to(DEV)))
          for shape in [(64, 32, 3, 3), (96, 64, 4), (40, 36, 3)]]
    opt = FlatAdam(ws, lr=1e-2)
    xs = [(2, 9, 11, 32), (2, 16, 64), (2, 13, 36)]

    def dx_all():
        out = []
        for w, xs_ in zip(ws, xs):
            x = torch.randn(xs_, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
            y = ops.conv_forward(x, w, None, 1, 1)
            out.append((ops.conv_input_grad(torch.ones_like(y), w, x.shape, 1, 1), w.detach().clone(), x.shape, y.shape))
        return out

    def check_all(tag):
        for dx, w, xshape, yshape in dx_all():
            conv_t = F.conv_transpose2d if w.dim() == 4 else F.conv_transpose1d
            gy = torch.ones(yshape, dtype=torch.float64).movedim(-1, 1)
            ref = conv_t(gy, w.double().cpu(), stride=1, padding=1).movedim(1, -1)
            check(tag, dx, ref, 1e-5)

    check_all("dX, fresh mirrors")
    assert not opt.mirrors.dirty and ops.WeightMirrors.lookup(ws[0]) is not None
    for w in ws:
        w.grad.copy_(torch.randn(w.shape, generator=g).to(DEV))
    opt.step()  # raw-pointer update of the weights
    assert opt.mirrors.dirty
    check_all("dX after an Adam step")
    with torch.no_grad():
        ws[1].copy_(torch.randn(ws[1].shape, generator=g).to(DEV) * 0.1)  # what load_state_dict does
    check_all("dX after an in-place edit")
    ptr = ws[0].data_ptr()
    del opt
    import gc
    gc.collect()
    assert ops.WeightMirrors.lookup(ws[0]) is None and ptr not in ops.WeightMirrors._by_ptr
    check_all("dX without mirrors")


def test_mel_frontend(ops):
    from oracle import sdt_oracle as O
    batch = O.make_batch(3, 16, step=0, seed=1)
    audio = batch["audio"]
    t = torch.arange(audio.shape[1]) / 16000.0
    audio[1] += 0.3 * torch.sin(2 * np.pi * (200 + 1500 * t) * t)  # chirp
    ref = O.mel_spectrogram(audio.double(), O.mel_window(torch.float64), O.mel_filterbank(torch.float64))
    basis = ops.dft_basis(O.mel_window()).to(DEV)
    mel = ops.mel_spectrogram(audio.to(DEV), basis, O.mel_filterbank().to(DEV))
    assert mel.shape == (3, 80, 427)
    check("mel vs float64 restatement", mel, ref, 2e-5)
    check("mel vs fp32 torch.stft oracle", mel, O.mel_spectrogram(audio), 2e-5)
    # odd length (demo path): F = 1 + L // 160
    a2 = audio[:, :50001]
    mel2 = ops.mel_spectrogram(a2.to(DEV), basis, O.mel_filterbank().to(DEV))
    check("mel L=50001", mel2, O.mel_spectrogram(a2.double(), O.mel_window(torch.float64), O.mel_filterbank(torch.float64)), 2e-5)


def test_time_diff(ops):
    x = torch.randn(3, 64, 242, dtype=torch.float64, requires_grad=True)
    y = x[:, 1:] - x[:, :-1]
    gy = torch.randn_like(y)
    y.backward(gy)
    xd = x.detach().float().to(DEV).requires_grad_(True)
    yd = ops.TimeDiffFn.apply(xd)
    yd.backward(gy.float().to(DEV))
    check("time diff fwd", yd, y, 1e-6)
    check("time diff bwd", xd.grad, x.grad, 1e-6)


def test_error_reporting(ops):
    # the C ABI returns a status + message; the binding raises (reference convention: Python exceptions)
    x = torch.zeros(2, 4, 6, device=DEV)
    with pytest.raises(RuntimeError, match="libsdt_hip"):
        ops.RowNormActFn.apply(x, 0.2)  # C=6 is not a multiple of 4


FULL_SIZE = [  # BASELINE config 2 sizes (B=32): name, Hi, Wi, Cin, Cout, kh, kw, s, p
    ("L1", 80, 427, 64, 64, 4, 4, 2, 1), ("L2", 40, 213, 64, 128, 3, 3, 1, 1), ("L5", 20, 106, 256, 256, 4, 4, 2, 1),
    ("L7", 10, 53, 256, 256, 6, 3, 1, 0), ("unet k4s2 T64", 1, 64, 256, 256, 1, 4, 2, 1),
]


@pytest.mark.parametrize("case", FULL_SIZE, ids=[c[0] for c in FULL_SIZE])
def test_conv_full_size_properties(ops, case):
    """At the benchmark's full sizes (B=32) the float64 reference is too slow, so the three conv kernels are tied together
    by size-independent identities of a linear map and its adjoints:
        <gy, conv(x, w)>  ==  <x, dX(gy, w)>  ==  <w, dW(x, gy)>           (forward, input-gradient, weight-gradient)
        conv(a*x1 + x2, w) == a*conv(x1, w) + conv(x2, w)                    (linearity)
    evaluated with float64 reductions of the fp32 kernel outputs."""
    tag, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    B = 32
    gen = torch.Generator(device=DEV).manual_seed(len(tag) * 7 + Cin)
    one_d = Hi == 1
    xs, ws = ((B, Wi, Cin), (Cout, Cin, kw)) if one_d else ((B, Hi, Wi, Cin), (Cout, Cin, kh, kw))
    x = torch.randn(xs, device=DEV, generator=gen)
    x2 = torch.randn(xs, device=DEV, generator=gen)
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(ws, device=DEV, generator=gen) * (2.0 / (Cin * kh * kw)) ** 0.5))
    y = ops.conv_forward(x, w, None, s, p)
    gy = torch.randn(y.shape, device=DEV, generator=gen)
    dx = ops.conv_input_grad(gy, w, x.shape, s, p)
    w.grad = None
    ops.conv_weight_grad(x, gy, w, s, p)
    torch.cuda.synchronize()
    a = (gy.double() * y.double()).sum().item()
    b = (x.double() * dx.double()).sum().item()
    c = (w.detach().double() * w.grad.double()).sum().item()
    scale = (gy.double().norm() * y.double().norm()).item()
    print("  %-16s <gy,y>=%.6e <x,dX>=%.6e <w,dW>=%.6e (|gy||y|=%.3e)" % (tag, a, b, c, scale))
    assert scale > 0 and abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    y12 = ops.conv_forward(0.5 * x + x2, w, None, s, p)
    y2 = ops.conv_forward(x2, w, None, s, p)
    check(tag + " linearity", y12, 0.5 * y.double() + y2.double(), 1e-5)  # three fp32 roundings of K ~ 4096 products


def test_flat_adam_resumes_from_a_torch_adam_state_dict(ops):
    """Checkpoint wire format (trainer.py:313-319): '<optimizer>_state_dict' written by the reference's torch.optim.Adam loads into
    FlatAdam (and back), and the next step agrees -- conv weights included, whose physical layout differs from the logical one."""
    from speechdrivestemplates_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(31)
    shapes = [(8, 6, 3, 3), (10, 8, 4), (10,), (5, 7)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g, dtype=torch.float64)) for s in shapes]
    ref_opt = torch.optim.Adam(ref_p, lr=1e-2)
    grads = [[torch.randn(s, generator=g, dtype=torch.float64) for s in shapes] for _ in range(3)]
    for p, gr in zip(ref_p, grads[0]):
        p.grad = gr.clone()
    ref_opt.step()  # the "checkpoint" is taken after one reference step
    sd = ref_opt.state_dict()
    mine_p = [torch.nn.Parameter(ops.to_weight_layout(p.detach().float().to(DEV)) if p.dim() >= 3 else p.detach().float().to(DEV)) for p in ref_p]
    opt = FlatAdam(mine_p, lr=1e-2)
    opt.load_state_dict(sd)
    assert int(opt.state_dev[0]) == 1
    for step in (1, 2):
        for p, q, gr in zip(ref_p, mine_p, grads[step]):
            p.grad = gr.clone()
            q.grad.copy_(gr.float().to(DEV))
        ref_opt.step()
        opt.step()
    for i, (p, q) in enumerate(zip(ref_p, mine_p)):
        check("param %d after resume + 2 steps" % i, q, p, 2e-6)
    back = opt.state_dict()  # and the other way: torch accepts what FlatAdam writes
    chk = torch.optim.Adam([torch.nn.Parameter(torch.zeros(s)) for s in shapes], lr=1e-2)
    chk.load_state_dict(back)
    assert float(chk.state_dict()["state"][0]["step"]) == 3.0
    check("exp_avg of the conv weight round-trips in logical layout", chk.state_dict()["state"][0]["exp_avg"], ref_opt.state_dict()["state"][0]["exp_avg"], 2e-6)


@pytest.mark.parametrize("norm,groups", [("IN", None), ("BN", 1)])
def test_conv_epilogue_statistics_match_the_separate_pass(ops, norm, groups):
    """sdt_conv_taps_stats_f32: the normalisation statistics accumulated by the conv epilogue give the same block output and
    gradients as conv -> colstats -> apply (tiles straddling two clips, ragged last tile included), and the float64 reference."""
    B, Hi, Wi, Cin, Cout = 3, 40, 61, 64, 128  # 3*2440 = 7320 rows: 114 full tiles + a ragged one, clip boundaries inside tiles
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (Cin * 9)) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, 1, 1)
    z = F.leaky_relu(F.instance_norm(y, eps=1e-5) if norm == "IN" else F.batch_norm(y, None, None, training=True, eps=1e-5), 0.2)
    gz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(gz)
    outs = []
    for fused in (True, False):
        ops.PROFILER_NO_FUSION = not fused
        try:
            xd = ops.cl(x.float()).to(DEV).requires_grad_(True)
            wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
            grp = B if norm == "IN" else 1
            assert ops.conv_stats_fusable(xd, wd, 1, 1, grp) == fused
            if fused:
                yd, sums = ops.ConvStatsFn.apply(xd, wd, 1, 1, grp)
                zd = ops.ColNormActFn.apply(yd, None, None, None, None, None, grp, 0.2, sums)
            else:
                zd = ops.ColNormActFn.apply(ops.ConvFn.apply(xd, wd, None, 1, 1), None, None, None, None, None, grp, 0.2)
            zd.backward(ops.cl(gz.float()).to(DEV))
            torch.cuda.synchronize()
            outs.append((ops.cf_view(zd).detach().clone(), ops.cf_view(xd.grad).clone(), wd.grad.clone()))
        finally:
            ops.PROFILER_NO_FUSION = False
    check(norm + " fused block fwd vs float64", outs[0][0], z, 2e-5)
    check(norm + " fused vs separate statistics", outs[0][0], outs[1][0], 2e-6)

    def l2(got, ref):  # an fp32 pre-activation within rounding of the LeakyReLU kink flips one derivative: compare in L2
        got, ref = got.double().cpu(), ref.double().cpu()
        return ((got - ref).norm() / ref.norm()).item()

    for name, i, ref in (("dX", 1, xr.grad), ("dW", 2, wr.grad)):
        e64, esep = l2(outs[0][i], ref), l2(outs[0][i], outs[1][i])
        print("  %s fused block %s: L2 error vs float64 %.2e, vs separate pass %.2e" % (norm, name, e64, esep))
        assert e64 < 3e-3 and esep < 1e-5, (name, e64, esep)  # the kink flips are shared by both fp32 paths


def test_tuning_environment_switches_are_ignored_by_the_product_library(tmp_path):
    """SDT_CONV_PRIO=3 used to select a "no global loads" ablation kernel (wrong results) inside libsdt_hip.so.  Those
    instantiations now live in the -DSDT_TUNING build only: with the variable set, the product library still convolves
    correctly."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from speechdrivestemplates_amd import ops
torch.manual_seed(0)
x = torch.randn(2, 20, 30, 64, device="cuda")
w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(64, 64, 3, 3, device="cuda") * 0.05))
y = ops.conv_forward(x, w, None, 1, 1)
ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.detach().double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
err = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
print("ERR", err)
assert err < 3e-6, err
''' % REPO
    env_lib = {k: v for k, v in os.environ.items() if k != "SDT_HIP_LIB"}  # the product library, whatever this session loaded
    for var in ("SDT_CONV_PRIO", "SDT_CONV_TILE", "SDT_STAGE1D", "SDT_PRESPLIT"):
        env = dict(env_lib, **{var: "3" if var == "SDT_CONV_PRIO" else ("128128" if var == "SDT_CONV_TILE" else "1")})
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (var, out.stdout[-2000:], out.stderr[-2000:])
    # no experiment switch is compiled into the product library or read by the package
    blob = open(os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip.so"), "rb").read()
    for needle in (b"SDT_CONV_PRIO", b"SDT_CONV_TILE", b"SDT_STAGE1D", b"SDT_PRESPLIT", b"c1d_kernel", b"conv_taps_pre_kernel", b"conv_tab_kernel", b"sk_dbg_mute"):
        assert needle not in blob, needle
    pkg = os.path.join(REPO, "speechdrivestemplates_amd")
    for root, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                for needle in ("SDT_STAGE1D", "SDT_PRESPLIT", "SDT_CONV_PRIO", "SDT_DETERMINISTIC_DW"):
                    assert ('environ.get("%s"' % needle) not in src and ("environ['%s']" % needle) not in src, (f, needle)


@pytest.mark.parametrize("norm,stride", [("IN", 2), ("IN", 1), ("BN", 2)])
def test_fused_dx_classes_and_backward_statistics(ops, norm, stride):
    """An input gradient is ONE launch over its output parity classes (sdt_conv_taps_multi_f32) and, in a sequential 2-D chain,
    its epilogue accumulates the statistics of the normalisation backward below it (sdt_norm_bwd) so that colstats<true> is
    skipped.  Both against float64 autograd of the same two-block chain, and against the unfused launches."""
    from speechdrivestemplates_amd.core.networks.building_blocks import ConvNormRelu
    torch.manual_seed(3)
    B, H, W = 3, 21, 50  # odd sizes: the parity classes have different extents
    blk1 = ConvNormRelu('2d', 32, 64, downsample=False, norm=norm, leaky=True).to(DEV).train()
    blk2 = ConvNormRelu('2d', 64, 32, downsample=(stride == 2), norm=norm, leaky=True).to(DEV).train()
    x = torch.randn(B, H, W, 32, device=DEV)
    gout = None
    res = {}
    for fused in (False, True):
        ops.FUSE_DX_CLASSES = ops.FUSE_BWD_STATS = fused
        try:
            for b in (blk1, blk2):
                for p in b.parameters():
                    p.grad = None
            xin = x.clone().requires_grad_(True)
            h1, h2 = ops.NormBwdHolder(), ops.NormBwdHolder()
            z1 = blk1.forward_cl(xin, None, h1)
            z2 = blk2.forward_cl(z1, h1, h2)
            if gout is None:
                gout = torch.randn_like(z2)
            before = dict(ops.HOLDER_HANDOVERS)
            z2.backward(gout)
            torch.cuda.synchronize()
            res[fused] = [xin.grad.clone()] + [p.grad.clone() for b in (blk1, blk2) for p in b.parameters()]
            if fused:  # block 1's normalisation backward took the sums that block 2's input-gradient launch accumulated (the stride-1
                # case is K-split at this size and does not fuse the statistics: nothing handed over, nothing refused)
                assert ops.HOLDER_HANDOVERS["used"] == before["used"] + (1 if stride == 2 else 0) and ops.HOLDER_HANDOVERS["refused"] == before["refused"]
        finally:
            ops.FUSE_DX_CLASSES = ops.FUSE_BWD_STATS = True
    for a, b in zip(res[True], res[False]):
        check("fused vs unfused launches", a, b, 2e-5)
    # the hand-over is refused when the gradient that reaches the normalisation is not the tensor the consumer's launch wrote (here: a
    # hook that rescales it) -- the statistics pass runs and the result is the gradient of the modified graph
    for b in (blk1, blk2):
        for p in b.parameters():
            p.grad = None
    xin = x.clone().requires_grad_(True)
    h1, h2 = ops.NormBwdHolder(), ops.NormBwdHolder()
    z1 = blk1.forward_cl(xin, None, h1)
    z1.register_hook(lambda g: g * 0.5)
    z2 = blk2.forward_cl(z1, h1, h2)
    before = dict(ops.HOLDER_HANDOVERS)
    z2.backward(gout)
    torch.cuda.synchronize()
    assert ops.HOLDER_HANDOVERS["refused"] == before["refused"] + (1 if stride == 2 else 0) and ops.HOLDER_HANDOVERS["used"] == before["used"]
    check("hooked gradient: dX is half the unhooked one", xin.grad, 0.5 * res[False][0], 2e-5)
    for got, ref in zip([p.grad for p in blk1.parameters()], res[False][1:]):
        check("hooked gradient: block-1 parameter gradients are half the unhooked ones", got, 0.5 * ref, 2e-5)
    # float64 reference of the chain
    def ref_block(blk, t):
        w = blk.conv.weight.detach().double().cpu().requires_grad_(True)
        y = F.conv2d(t, w, None, blk.stride, blk.padding)
        if norm == "IN":
            y = F.instance_norm(y, eps=1e-5)
            extra = []
        else:
            ga = blk.norm.weight.detach().double().cpu().requires_grad_(True)
            be = blk.norm.bias.detach().double().cpu().requires_grad_(True)
            y = F.batch_norm(y, None, None, ga, be, True, 0.1, 1e-5)
            extra = [ga, be]
        return F.leaky_relu(y, 0.2), [w] + extra
    xr = x.detach().double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    z1r, p1 = ref_block(blk1, xr)
    z2r, p2 = ref_block(blk2, z1r)
    z2r.backward(gout.double().cpu().permute(0, 3, 1, 2))
    check("chain dX vs float64", res[True][0], xr.grad.permute(0, 2, 3, 1), 2e-4)
    for got, ref in zip(res[True][1:], [p.grad for p in p1 + p2]):
        check("chain parameter gradient vs float64", got, ref, 5e-4)


@pytest.mark.parametrize("case", [CONV2D[1], CONV2D[5], CONV2D[7], ("1-D 256->256 k4s2", 4, 1, 64, 256, 256, 1, 4, 2, 1),
                                  ("ragged 3->5 k3", 2, 9, 11, 3, 5, 3, 3, 1, 1)], ids=lambda c: c[0])
def test_deterministic_weight_gradient(ops, case):
    """ops.DETERMINISTIC_DW (sdt_conv_dw_det_f32): row-range slabs + an ordered reduce instead of fp32 atomics.  Repeated launches
    are BIT-identical (the atomic path is not: it is checked to differ by rounding only), the result equals the float64 weight
    gradient to the tolerance of the atomic path, and it accumulates into an existing gradient like the atomic path does."""
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    one_d = Hi == 1
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    if one_d:
        x = torch.randn(B, Cin, Wi, generator=g, dtype=torch.float64)
        w = torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) * (2.0 / (Cin * kw)) ** 0.5
        conv = F.conv1d
    else:
        x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
        w = torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) * (2.0 / (Cin * kh * kw)) ** 0.5
        conv = F.conv2d
    wr = w.clone().requires_grad_(True)
    y = conv(x, wr, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = ops.cl(x.float()).to(DEV), ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    ref = ops.to_weight_layout(wr.grad)
    prev = ops.DETERMINISTIC_DW
    try:
        ops.DETERMINISTIC_DW = True
        runs = []
        for _ in range(3):
            wd.grad = None
            ops.conv_weight_grad(xd, gyd, wd, s, p)
            torch.cuda.synchronize()
            runs.append(wd.grad.clone())
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), "deterministic weight gradient differs between runs"
        check(tag + " dW (deterministic)", runs[0], ref, 5e-5)
        ops.conv_weight_grad(xd, gyd, wd, s, p)  # accumulates
        check(tag + " dW (deterministic, accumulated)", wd.grad, 2 * ref, 5e-5)
        ops.DETERMINISTIC_DW = False
        wd.grad = None
        ops.conv_weight_grad(xd, gyd, wd, s, p)
        torch.cuda.synchronize()
        check(tag + " dW (atomics) vs deterministic", wd.grad, runs[0].double(), 2e-5)
    finally:
        ops.DETERMINISTIC_DW = prev


def test_lsgan_mse_terms(ops):
    """ops.MseConstFn (sdt_mse_const_*_f32) = nn.MSELoss(scores, full_like(scores, target)) * lambda, forward and gradient
    (voice2pose.py:189-197)."""
    g = torch.Generator().manual_seed(5)
    for shape, target, lam in (((32, 1, 7), 1.0, 1.0), ((4, 1, 3), 0.0, 0.5), ((3, 1000), 1.0, 2.0)):
        s = torch.randn(shape, generator=g, dtype=torch.float64)
        sr = s.clone().requires_grad_(True)
        ref = F.mse_loss(sr, torch.full_like(sr, target)) * lam
        (ref * 1.5).backward()
        sd = s.float().to(DEV).requires_grad_(True)
        out = ops.MseConstFn.apply(sd, target, lam)
        (out * 1.5).backward()
        torch.cuda.synchronize()
        check("lsgan term %s" % (shape,), out.reshape(1), ref.detach().reshape(1), 1e-6)
        check("lsgan term gradient %s" % (shape,), sd.grad, sr.grad, 1e-6)


# (tag, B, Hi, Wi, Cin, Cout, k, s, p): one case per tile shape of convsk_dw_kernel (rows x columns of the dW tile)
SK_DW = [("128x128 tile", 8, 20, 106, 128, 256, 3, 1, 1), ("128x64 tile", 4, 40, 213, 64, 128, 3, 1, 1), ("64x128 tile", 4, 80, 427, 64, 64, 4, 2, 1),
         ("64x64 tile", 4, 40, 213, 64, 64, 3, 1, 1), ("128x64 tile, Cout 192", 4, 20, 106, 64, 192, 3, 1, 1), ("64x128 tile, ragged rows", 5, 33, 77, 192, 64, 3, 1, 1)]


def _dw_tile_rule(case, wide, grid=512):
    """csrc/convsk.hip dw_tile restated: rows by the divisibility of Cout, columns by that of taps * Cin; the ragged-tile rule (ops.SK_DW_WIDE) widens
    64-column tiles to 128 (the 576-column cases: five tiles with a ragged last one) unless that leaves the K loop fewer than 8 steps per chunk."""
    _tag, B, Hi, Wi, Cin, Cout, k, s, p = case
    Ho, Wo = (Hi + 2 * p - k) // s + 1, (Wi + 2 * p - k) // s + 1
    N, K = k * k * Cin, -(-(B * Ho * Wo) // 32)
    bm, bn = (128 if Cout % 128 == 0 else 64), (128 if N % 128 == 0 else 64)
    if wide and bn == 64 and N >= 128:
        T = (Cout // bm) * -(-N // 128)
        if T <= grid and K >= 8 * (grid // T):
            bn = 128
    return bm, bn


@pytest.mark.parametrize("wide", [True, False], ids=["wide-tiles", "r5-tiles"])
@pytest.mark.parametrize("case", SK_DW, ids=lambda c: c[0])
def test_streamk_weight_gradient_tile_shapes(ops, case, wide):
    """sdt_convsk_dw_f32 (ordered (tile, K-chunk) units + fixed-order slab reduce) on every tile shape it instantiates: the launch takes
    that kernel (sdt_convsk_dw_supported), equals the float64 weight gradient, repeats bit-identically and accumulates.  Round 6: both tile rules of
    the split-fp32 kernel (the case names are the rule of rounds 3-5)."""
    prev, ops.SK_DW_WIDE = ops.SK_DW_WIDE, wide
    try:
        _dw_tile_case(ops, case, wide)
    finally:
        ops.SK_DW_WIDE = prev


def _dw_tile_case(ops, case, wide):
    from speechdrivestemplates_amd import _lib
    tag, B, Hi, Wi, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) * (2.0 / (Cin * k * k)) ** 0.5
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(x, wr, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = ops.cl(x.float()).to(DEV), ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    geom = ops.conv_geom_for(xd.shape, wd, s, p)
    assert ops.USE_STREAMK_DW and _lib.load().sdt_convsk_dw_supported(geom), "this case must run on the stream-K weight-gradient kernel"
    if ops.F32_SPLIT:
        plan = ops._sk_dw_plan(geom, xd.device)
        assert plan is not None and (plan.host[3] >> 26) & 1
        want = _dw_tile_rule(case, wide)
        assert (plan.host[1], plan.host[2]) == want, (tag, wide, plan.host[1], plan.host[2], want)
    runs = []
    for _ in range(2):
        wd.grad = None
        ops.conv_weight_grad(xd, gyd, wd, s, p)
        torch.cuda.synchronize()
        runs.append(wd.grad.clone())
    assert torch.equal(runs[0], runs[1]), "stream-K weight gradient differs between runs"
    check("streamk dW " + tag, runs[0], wr.grad, 5e-6)
    ops.conv_weight_grad(xd, gyd, wd, s, p)
    check("streamk dW accumulated " + tag, wd.grad, 2 * wr.grad, 5e-6)
    assert ops.streamk_error_codes() == {}, ops.streamk_error_codes()


@pytest.mark.parametrize("case", [("(6,3) valid", 16, 10, 53, 256, 256, 6, 3, 1, 0), ("3x3 pad 1", 16, 10, 53, 256, 256, 3, 3, 1, 1)], ids=lambda c: c[0])
@pytest.mark.parametrize("groups", ["IN", "BN"])
def test_streamk_row_major_tile_order(ops, case, groups):
    """Launches whose plan orders the GEMM rows image-row-major ((oy, b, ox): convsk.hip plan_build -- short images, >= 64 row tiles, >= 5 %
    of the K steps culled): forward (+ statistics epilogue), input gradient and its normalisation-backward sums against float64."""
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    ng = B if groups == "IN" else 1
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) * (2.0 / (Cin * kh * kw)) ** 0.5
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(xr, w, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = ops.cl(x.float()).to(DEV), ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    ops.begin_step()
    yd, sums = ops.ConvStatsFn.apply(xd, wd, s, p, ng)
    torch.cuda.synchronize()
    check("row-major fwd %s %s" % (tag, groups), ops.cf_view(yd), y, 3e-6)
    yg = y.permute(0, 2, 3, 1).reshape(ng, -1, Cout)
    check("row-major fwd statistics %s %s" % (tag, groups), sums.view(ng, Cout, 2), torch.stack([yg.sum(1), (yg * yg).sum(1)], -1), 3e-6)
    # input gradient with the statistics of the normalisation that produced x: sum gg, sum gg * yhat with gg = dx * act'(gamma * yhat + beta)
    h = ops.NormBwdHolder()
    h.y = ops.cl(torch.randn(x.shape, generator=g)).to(DEV)
    h.mean = (torch.randn(ng, Cin, generator=g) * 0.1).to(DEV)
    h.rstd = (torch.rand(ng, Cin, generator=g) + 0.5).to(DEV)
    h.gamma, h.beta = (torch.rand(Cin, generator=g) + 0.5).to(DEV), (torch.randn(Cin, generator=g) * 0.1).to(DEV)
    h.groups, h.slope = ng, 0.2
    dx = ops.conv_input_grad(gyd, wd, xd.shape, s, p, h)
    torch.cuda.synchronize()
    check("row-major dX %s %s" % (tag, groups), ops.cf_view(dx), xr.grad, 3e-6)
    assert h.sums is not None, "the input-gradient launch did not fuse the backward statistics"
    dxg = xr.grad.permute(0, 2, 3, 1).reshape(ng, -1, Cin)
    yh = (h.y.double().cpu().reshape(ng, -1, Cin) - h.mean.double().cpu()[:, None]) * h.rstd.double().cpu()[:, None]
    pre = yh * h.gamma.double().cpu() + h.beta.double().cpu()
    gg = dxg * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, 0.2))
    check("row-major dX statistics %s %s" % (tag, groups), h.sums.view(ng, Cin, 2), torch.stack([gg.sum(1), (gg * yh).sum(1)], -1), 3e-6)
    assert ops.streamk_error_codes() == {}, ops.streamk_error_codes()


def test_statistics_sums_do_not_depend_on_the_arrival_order(ops):
    """The normalisation statistics are fp64 atomics over per-32-row-block fp32 partial sums.  Such a sum is exact in fp64 (24-bit terms of similar
    magnitude, <= 2^10 of them), hence independent of the order in which the workgroups arrive (docs/DESIGN_rounds_1-5.md section 2): repeated launches of the forward
    and the backward statistics epilogues give BITWISE identical fp64 sums."""
    torch.manual_seed(5)
    B = 16
    for (Hi, Wi, Cin, Cout, k, s, p) in [(20, 106, 128, 256, 3, 1, 1), (40, 213, 64, 128, 3, 1, 1)]:
        x = torch.randn((B, Hi, Wi, Cin), device=DEV)
        w = torch.nn.Parameter(ops.to_weight_layout(torch.randn((Cout, Cin, k, k), device=DEV) * 0.05))
        first = None
        for rep in range(6):
            ops.begin_step()
            y, sums = ops.ConvStatsFn.apply(x, w, s, p, B)
            if rep == 0:
                gy = torch.randn_like(y)
                hy, hm, hr = torch.randn_like(x), torch.randn((B, Cin), device=DEV) * 0.1, torch.rand((B, Cin), device=DEV) + 0.5
            h = ops.NormBwdHolder()
            h.y, h.mean, h.rstd, h.gamma, h.beta, h.groups, h.slope = hy, hm, hr, None, None, B, 0.2
            ops.conv_input_grad(gy, w, x.shape, s, p, h)
            torch.cuda.synchronize()
            assert h.sums is not None
            now = (sums.clone().view(torch.int64), h.sums.clone().view(torch.int64))
            if first is None:
                first = now
            else:
                assert torch.equal(now[0], first[0]), "forward statistics differ between launches"
                assert torch.equal(now[1], first[1]), "backward statistics differ between launches"


def test_streamk_plans_with_reserved_slots(ops):
    """Data-parallel runs plan their backward stream-K launches with workgroup slots left free for the collective's kernels
    (ops.SK_RESERVED_SLOTS, set by dp.GradReducer): the grid shrinks, forward plans keep the whole GPU, results stay within the fp32 bars."""
    tag, B, Hi, Wi, Cin, Cout, k, s, p = ("reserve", 16, 20, 106, 128, 256, 3, 1, 1)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) * (2.0 / (Cin * k * k)) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = ops.cl(x.float()).to(DEV), ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    prev = ops.SK_RESERVED_SLOTS
    ops.clear_plans()  # (the cache is process-wide: other tests' plans of other routing settings would be counted below)
    try:
        ops.SK_RESERVED_SLOTS = 32
        yd = ops.conv_forward(xd, wd, None, s, p)
        dx = ops.conv_input_grad(gyd, wd, xd.shape, s, p)
        ops.conv_weight_grad(xd, gyd, wd, s, p)
        torch.cuda.synchronize()
        # plan cache keys: (geometry bytes, classes, rows per group, backward groups, device, forward, reserve, dtype, routing knobs ...)
        grids = {(key[5], key[6]): plan.host[3] & 0xffff for key, plan in ops._SK_PLANS.items() if plan is not None and key[6] == 32}
        dw_grids = [plan.host[3] & 0xffff for key, plan in ops._SK_DW_PLANS.items() if plan is not None and key[2] == 32]
    finally:
        ops.SK_RESERVED_SLOTS = prev
    # backward plans: 512 - 32 workgroups (two 4-wave workgroups per CU), or 256 - 16 one-per-CU 8-wave workgroups (the split-fp32 kernel: a slot is half a CU)
    per_cu = 1 if ops.F32_SPLIT else 2
    assert grids and all(v == 256 * per_cu - 16 * per_cu for v in grids.values()), grids
    assert dw_grids and all(v == 480 for v in dw_grids), dw_grids
    # (fp32 plans: key[7] is the element type -- bf16 plans of other tests in this process may be the one-workgroup-per-CU kind, 256 ranges)
    fwd = [plan.host[3] & 0xffff for key, plan in ops._SK_PLANS.items() if plan is not None and key[5] and key[6] == 0 and key[7] == 0]
    assert fwd and all(v == 256 * per_cu for v in fwd), fwd              # forward plans keep every slot
    check("reserve fwd", ops.cf_view(yd), y, 3e-6)
    check("reserve dX", ops.cf_view(dx), xr.grad, 3e-6)
    check("reserve dW", wd.grad, wr.grad, 5e-6)
    assert ops.streamk_error_codes() == {}, ops.streamk_error_codes()


def _random_conv_cases(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    while len(out) < n:
        B = int(rng.choice([1, 2, 3, 5, 8, 17]))
        Cin, Cout = int(rng.choice([64, 128, 192, 256])), int(rng.choice([64, 128, 192, 256]))
        kh, kw, s, p = [(3, 3, 1, 1), (4, 4, 2, 1), (6, 3, 1, 0), (1, 1, 1, 0), (3, 3, 1, 0), (3, 3, 2, 1), (4, 4, 2, 0)][int(rng.integers(7))]  # (the tap table holds 20 taps: the reference stops at 6x3)
        Hi, Wi = int(rng.integers(max(kh, 6), 48)), int(rng.integers(max(kw, 33), 120))
        if (Hi + 2 * p - kh) // s + 1 < 1 or (Wi + 2 * p - kw) // s + 1 < 1:
            continue
        out.append(("B%d %dx%d %d->%d k%dx%d s%d p%d" % (B, Hi, Wi, Cin, Cout, kh, kw, s, p), B, Hi, Wi, Cin, Cout, kh, kw, s, p))
    return out


@pytest.mark.parametrize("case", _random_conv_cases(24, 2026), ids=lambda c: c[0])
def test_conv_random_shapes_through_the_production_routing(ops, case):
    """Seeded sweep of awkward shapes (odd images, ragged row counts, 1..17 clips, all channel widths the kernels tile differently, strides,
    valid / padded / 1x1 kernels) through ops.conv_forward / conv_input_grad / conv_weight_grad exactly as the model calls them -- whichever
    kernel the routing picks (stream-K plans where they exist and have enough work, the 64x64 kernels otherwise) -- against float64."""
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) * (2.0 / (Cin * kh * kw)) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = ops.cl(x.float()).to(DEV), ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    yd = ops.conv_forward(xd, wd, None, s, p)
    dx = ops.conv_input_grad(gyd, wd, xd.shape, s, p)
    ops.conv_weight_grad(xd, gyd, wd, s, p)
    torch.cuda.synchronize()
    check("random fwd " + tag, ops.cf_view(yd), y, 4e-6)
    check("random dX " + tag, ops.cf_view(dx), xr.grad, 4e-6)
    check("random dW " + tag, wd.grad, wr.grad, 8e-6)
    assert ops.streamk_error_codes() == {}, ops.streamk_error_codes()


def _random_conv1d_cases(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    while len(out) < n:
        B = int(rng.choice([1, 2, 4, 7, 32]))
        Cin, Cout = int(rng.choice([32, 64, 256, 288])), int(rng.choice([64, 128, 256, 242]))
        k, s, p = [(3, 1, 1), (4, 2, 1), (1, 1, 0), (3, 1, 0), (4, 2, 0), (2, 2, 0)][int(rng.integers(6))]
        T = int(rng.integers(max(k, 1), 70))
        if (T + 2 * p - k) // s + 1 < 1:
            continue
        out.append(("B%d T%d %d->%d k%d s%d p%d" % (B, T, Cin, Cout, k, s, p), B, T, Cin, Cout, k, s, p))
    return out


@pytest.mark.parametrize("case", _random_conv1d_cases(20, 7), ids=lambda c: c[0])
def test_conv1d_random_shapes_through_the_production_routing(ops, case):
    """The 1-D stage's launches on a seeded sweep of lengths (1..69 frames), clip counts, strides and channel widths -- including the widths the
    small-K kernel does not take (Cout = 242) -- with and without bias, against float64: conv1d_small_kernel / conv_taps_kernel, K-split or not,
    two parity classes for the strided input gradients."""
    tag, B, T, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    x = torch.randn(B, Cin, T, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, generator=g, dtype=torch.float64) * (2.0 / (Cin * k)) ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64) if (len(tag) % 2) else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv1d(xr, wr, bias, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = ops.cl(x.float()).to(DEV), ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    bd = bias.float().to(DEV) if bias is not None else None
    yd = ops.conv_forward(xd, wd, bd, s, p)
    dx = ops.conv_input_grad(gyd, wd, xd.shape, s, p)
    ops.conv_weight_grad(xd, gyd, wd, s, p)
    torch.cuda.synchronize()
    check("random 1-D fwd " + tag, ops.cf_view(yd), y, 4e-6)
    check("random 1-D dX " + tag, ops.cf_view(dx), xr.grad, 4e-6)
    check("random 1-D dW " + tag, wd.grad, wr.grad, 1e-5)


def _random_stats_cases(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    while len(out) < n:
        B = int(rng.choice([2, 3, 8, 16, 32]))
        Cin, Cout = int(rng.choice([64, 128, 256])), int(rng.choice([64, 128, 256]))
        kh, kw, s, p = [(3, 3, 1, 1), (4, 4, 2, 1), (6, 3, 1, 0), (3, 3, 1, 0)][int(rng.integers(4))]
        Hi, Wi = int(rng.integers(max(kh, 6), 44)), int(rng.integers(max(kw, 40), 110))
        norm = "IN" if rng.integers(2) else "BN"
        out.append(("%s B%d %dx%d %d->%d k%dx%d s%d p%d" % (norm, B, Hi, Wi, Cin, Cout, kh, kw, s, p), norm, B, Hi, Wi, Cin, Cout, kh, kw, s, p))
    return out


@pytest.mark.parametrize("case", _random_stats_cases(14, 99), ids=lambda c: c[0])
def test_statistics_epilogues_random_shapes(ops, case):
    """Forward statistics (sum, sum of squares per group and channel) and normalisation-backward statistics (sum gg, sum gg * yhat) out of the
    conv epilogues -- whichever kernel and row order the routing picks, InstanceNorm (one group per clip) or BatchNorm (one group) -- on a seeded
    sweep of shapes against float64.  Where a launch does not fuse the statistics (too few rows, K-split) only the conv results are checked."""
    tag, norm, B, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    ng = B if norm == "IN" else 1
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) * (2.0 / (Cin * kh * kw)) ** 0.5
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(xr, w, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = ops.cl(x.float()).to(DEV), ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    ops.begin_step()
    if ops.conv_stats_fusable(xd, wd, s, p, ng):
        yd, sums = ops.ConvStatsFn.apply(xd, wd, s, p, ng)
        torch.cuda.synchronize()
        yg = y.permute(0, 2, 3, 1).reshape(ng, -1, Cout)
        check("random forward statistics " + tag, sums.view(ng, Cout, 2), torch.stack([yg.sum(1), (yg * yg).sum(1)], -1), 4e-6)
    else:
        yd = ops.conv_forward(xd, wd, None, s, p)
    check("random stats fwd " + tag, ops.cf_view(yd), y, 4e-6)
    h = ops.NormBwdHolder()
    h.y = ops.cl(torch.randn(x.shape, generator=g)).to(DEV)
    h.mean = (torch.randn(ng, Cin, generator=g) * 0.1).to(DEV)
    h.rstd = (torch.rand(ng, Cin, generator=g) + 0.5).to(DEV)
    h.gamma, h.beta = (torch.rand(Cin, generator=g) + 0.5).to(DEV), (torch.randn(Cin, generator=g) * 0.1).to(DEV)
    h.groups, h.slope = ng, 0.2
    dx = ops.conv_input_grad(gyd, wd, xd.shape, s, p, h)
    torch.cuda.synchronize()
    check("random stats dX " + tag, ops.cf_view(dx), xr.grad, 4e-6)
    if h.sums is not None:
        dxg = xr.grad.permute(0, 2, 3, 1).reshape(ng, -1, Cin)
        yh = (h.y.double().cpu().reshape(ng, -1, Cin) - h.mean.double().cpu()[:, None]) * h.rstd.double().cpu()[:, None]
        pre = yh * h.gamma.double().cpu() + h.beta.double().cpu()
        gg = dxg * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, 0.2))
        check("random backward statistics " + tag, h.sums.view(ng, Cin, 2), torch.stack([gg.sum(1), (gg * yh).sum(1)], -1), 4e-6)
    assert ops.streamk_error_codes() == {}, ops.streamk_error_codes()


def test_gradients_are_ready_when_backward_returns(ops):
    """Weight-gradient launches go to a side stream; autograd's contract -- .grad is ready on the current stream when backward() returns -- is
    kept by joining that stream in an engine callback at the end of the pass: a consumer on the main stream right after backward() (a plain torch
    optimiser, a .cpu() copy) sees the finished gradient.  Repeated to give a race a chance to show."""
    torch.manual_seed(2)
    B, H, W, Cin, Cout = 8, 40, 213, 64, 128  # 10 GFLOP: above ops.OVERLAP_DW_MIN_FLOPS, the weight gradient takes the side stream
    x = torch.randn(B, H, W, Cin, device=DEV)
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05))
    gy = torch.randn(B, H, W, Cout, device=DEV)
    ref = None
    for rep in range(6):
        w.grad = None
        y = ops.ConvFn.apply(x, w, None, 1, 1)
        y.backward(gy)
        got = w.grad.clone()          # main stream, immediately after backward()
        if ref is None:
            torch.cuda.synchronize()
            ref = w.grad.clone()
            assert float(ref.abs().max()) > 0
        assert torch.equal(got, ref), "repetition %d read an unfinished weight gradient" % rep


def _streamk_launch(ops, seed=3):
    """one persistent stream-K forward launch with split tiles (B=16 of the L4 shape); returns (y, float64 reference)"""
    g = torch.Generator().manual_seed(seed)
    B, Hi, Wi, Cin, Cout, k = 16, 20, 106, 128, 256, 3
    x = torch.randn(B, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) * (2.0 / (Cin * k * k)) ** 0.5
    xd = ops.cl(x.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    y = ops.conv_forward(xd, wd, None, 1, 1)
    torch.cuda.synchronize()
    return y, F.conv2d(x, w, None, 1, 1)


def test_streamk_error_word_makes_the_trainer_raise(ops):
    """The error word of a stream-K workspace (a lost partner, see the next test) is what Trainer.check_kernels -- called on log steps, after
    validation and before every checkpoint -- turns into an exception: training never logs or saves numbers such a launch produced."""
    from speechdrivestemplates_amd.core.pipelines.trainer import Trainer
    y, ref = _streamk_launch(ops)
    check("stream-K forward before the simulated failure", ops.cf_view(y), ref, 3e-6)
    assert ops.streamk_error_codes() == {}
    Trainer.check_kernels()  # clean: no exception
    key, ws = next(iter(ops._SK_WS.items()))
    ws[ops._SK_ERR_WORD] = 7  # what the kernel stores when range 6 never raised its flag
    try:
        assert ops.streamk_error_codes() == {key: 7}
        with pytest.raises(RuntimeError, match="gave up waiting for a partner"):
            Trainer.check_kernels()
        with pytest.raises(RuntimeError, match="gave up waiting for a partner"):
            ops.check_streamk()
    finally:
        ws[ops._SK_ERR_WORD] = 0
    Trainer.check_kernels()


@pytest.mark.tuning
@pytest.mark.parametrize("variant", ["split-f32 (convbf2_kernel<float>)", "fp32 MFMA (convsk_kernel)", "bf16 tensors (convbf2_kernel<bf16>)"])
def test_streamk_lost_partner_is_loud(ops, variant):
    """Fault injection (-DSDT_TUNING library): one workgroup of a stream-K launch computes its partial tile but never raises its flag -- a
    partner that was never dispatched.  The owner gives up after the spin limit, sets the error word FIRST and then stores the tile as NaN (never a
    tile with a partial sum missing; never NaN with a clean word); the next, healthy launch on the same workspace is correct
    again.  Round 6: one case per persistent forward / input-gradient kernel family -- each has its own give-up path."""
    from speechdrivestemplates_amd import _lib
    lib = _lib.load()
    prev = lib.sdt_convsk_get_spin_limit()
    prev_split = ops.F32_SPLIT
    ops.F32_SPLIT = not variant.startswith("fp32 MFMA")
    bf16 = variant.startswith("bf16")

    def launch(seed=3):
        if not bf16:
            return _streamk_launch(ops, seed)
        g = torch.Generator().manual_seed(seed)
        B, Hi, Wi, Cin, Cout, k = 16, 20, 106, 128, 256, 3
        x = torch.randn(B, Cin, Hi, Wi, generator=g).to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(torch.bfloat16)
        xd = ops.cl(x).to(DEV)
        wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
        y, _sums = ops.ConvStatsFn.apply(xd, wd, 1, 1, B, None)  # bf16 input -> the bf16-storage forward launch (statistics epilogue)
        torch.cuda.synchronize()
        return y.float(), F.conv2d(x.double(), w.double(), None, 1, 1)

    tol = 2e-2 if bf16 else 3e-6  # (bf16: the OUTPUT is stored as bf16)
    _lib.check(lib.sdt_convsk_set_spin_limit(20000))
    try:
        # which ranges are partners (not owners) of a split tile depends on the plan -- tile shape, workgroups, reserve: whatever ran before in this
        # process -- so a few candidates are tried: muting a range that owns every tile it touches changes nothing (and must change nothing)
        bad = None
        for cand in ([100, 101, 102, 103, 57, 58, 59] if bf16 else [200, 201, 202, 203, 57, 58, 59]):
            _lib.check(lib.sdt_debug_convsk_mute_range(cand))
            y, ref = launch()
            bad = torch.isnan(y)
            if bad.any():
                break
            assert ops.streamk_error_codes() == {}, ("a muted OWNER range cannot lose anything", cand, ops.streamk_error_codes())
        codes = ops.streamk_error_codes()
        assert bad.any(), "none of the muted ranges was a partner of a split tile: pick other ranges for this plan"
        assert len(codes) == 1 and list(codes.values())[0] > 0, ("NaN tile with a clean error word", codes)  # NaN => word set
        assert bad.sum().item() <= 256 * 256, "exactly one tile is poisoned"
        ok = ~bad.cpu()
        err = ((ops.cf_view(y).double().cpu() - ref).abs()[ops.cf_view(ok)]).max().item() / ref.abs().max().item()
        assert err < tol, err  # every other tile is right
        with pytest.raises(RuntimeError, match="gave up waiting for a partner"):
            ops.check_streamk()
    finally:
        _lib.check(lib.sdt_debug_convsk_mute_range(-1))
        _lib.check(lib.sdt_convsk_set_spin_limit(prev))
        for ws in ops._SK_WS.values():
            ws[ops._SK_ERR_WORD] = 0
    try:
        y, ref = launch(seed=4)
        e = ((ops.cf_view(y).double().cpu() - ref).abs().max() / ref.abs().max()).item()
        assert torch.isfinite(y).all() and e < tol, e
        assert ops.streamk_error_codes() == {}
    finally:
        ops.F32_SPLIT = prev_split


def test_input_gradient_with_unreachable_rows(ops):
    """k5 / stride 2 / no padding over 10 x 65 inputs: the last input row of every image is reached by no output position, whole tiles of a parity
    class have no live K step, the stream-K plan builder refuses the pack (tests/test_geometry.py) and the 64x64 kernel computes the gradient --
    zeros included -- against float64."""
    B, Hi, Wi, Cin, Cout, k, s, p = 32, 10, 65, 128, 128, 5, 2, 0
    g = torch.Generator().manual_seed(23)
    x = torch.randn(2, Cin, Hi, Wi, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) * 0.02
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(xr, w, None, s, p)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    gyd = torch.zeros((B,) + tuple(y.shape[2:]) + (Cout,), device=DEV)
    gyd[:2] = ops.cl(gy.float()).to(DEV)
    wd = torch.nn.Parameter(ops.to_weight_layout(w.float()).to(DEV))
    arr, n, _gs = ops.dx_pack(B, Hi, Wi, Cin, Cout, k, k, s, p, False)
    assert ops._sk_plan(arr, n, -1, 1, gyd.device) is None
    dx = torch.full((B, Hi, Wi, Cin), float("nan"), device=DEV)  # poison: torch.empty inside the op could hide an unwritten row behind stale zeros
    del dx
    dx = ops.conv_input_grad(gyd, wd, (B, Hi, Wi, Cin), s, p)
    torch.cuda.synchronize()
    assert torch.isfinite(dx).all()
    check("k5 s2 dX (items 0, 1)", ops.cf_view(dx[:2]), xr.grad, 3e-6)
    assert float(dx[2:].abs().max()) == 0.0 and float(dx[:, Hi - 1].abs().max()) <= float(xr.grad[:, :, Hi - 1].abs().max()) + 1e-6


# (tag, Hi, Wi, Cin, Cout, k, stride, pad, B): the last three have 530 / 255 output rows per clip -- 128-row tiles straddle clips (two statistics groups in a tile,
# partial last tile), odd batch sizes
SPLIT_CASES = [("L1", 80, 427, 64, 64, 4, 2, 1, 8), ("L2", 40, 213, 64, 128, 3, 1, 1, 8), ("L3", 40, 213, 128, 128, 4, 2, 1, 8), ("L4", 20, 106, 128, 256, 3, 1, 1, 8),
               ("L5", 20, 106, 256, 256, 4, 2, 1, 23), ("L6", 10, 53, 256, 256, 3, 1, 1, 27), ("L7", 10, 53, 256, 256, (6, 3), 1, 0, 49)]


@pytest.mark.parametrize("case", SPLIT_CASES, ids=[c[0] for c in SPLIT_CASES])
def test_split_f32_conv_vs_float64_and_the_fp32_mfma_kernels(ops, case):
    """The split-fp32 form of the 8-wave conv kernel (csrc/convbf.hip, ET = float: each fp32 operand = three bf16 numbers, six bf16 MFMA products per
    fp32 product, chunked fp32 accumulation) against float64 (building_blocks.py:15-22's Conv2d): forward with statistics, plain forward, input
    gradient.  Bar: the RMS error against float64 is at most 1.25 x that of the fp32-MFMA kernels of rounds 3-4 on the same tensors (measured: 0.6-1.0 x,
    tools/debug/x3_check.py), the maximum error inside the per-layer fp32 bar of the other conv tests (3e-6 of the output's maximum), and the
    (clip, channel) sums / sums of squares within 2e-7 of float64's."""
    tag, Hi, Wi, Cin, Cout, k, s, p, B = case
    kh, kw = k if isinstance(k, tuple) else (k, k)
    g = torch.Generator().manual_seed(3 + Hi)
    x = torch.randn(B, Hi, Wi, Cin, generator=g)
    wl = torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wl.double(), None, s, p).permute(0, 2, 3, 1).contiguous()
    gy = torch.randn(ref.shape, generator=g)
    ref_dx = torch.nn.grad.conv2d_input((B, Cin, Hi, Wi), wl.double(), gy.permute(0, 3, 1, 2).double(), s, p).permute(0, 2, 3, 1).contiguous()
    rs = torch.stack([ref.reshape(B, -1, Cout).sum(1), ref.reshape(B, -1, Cout).pow(2).sum(1)], -1).reshape(-1)
    xd, gyd = x.to(DEV), gy.to(DEV)
    w = torch.nn.Parameter(ops.to_weight_layout(wl).to(DEV))
    dev = torch.device(DEV, 0)
    prev = ops.F32_SPLIT
    out = {}
    try:
        for split in (False, True):
            ops.F32_SPLIT = split
            ops.clear_plans()
            ops._ARENA.begin_step(dev)
            y, sums = ops.ConvStatsFn.apply(xd, w, s, p, B, None)
            y0 = ops.conv_forward(xd, w, None, s, p)
            dx = ops.conv_input_grad(gyd, w, xd.shape, s, p, None)
            torch.cuda.synchronize()
            kinds = {(pl.host[3] >> 26) & 1 for pl in ops._SK_PLANS.values() if pl is not None}
            assert kinds == ({1} if split else {0}), (split, kinds)  # the launches really took the kernel under test
            out[split] = (y.double().cpu(), y0.double().cpu(), dx.double().cpu(), sums.double().cpu().reshape(-1).clone())
    finally:
        ops.F32_SPLIT = prev
        ops.clear_plans()

    def rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

    for i, (name, r) in enumerate((("forward + statistics", ref), ("forward", ref), ("input gradient", ref_dx))):
        e_split, e_mfma = rms(out[True][i], r), rms(out[False][i], r)
        print("  %s %-22s rms error vs float64: split %.3e  fp32 MFMA %.3e" % (tag, name, e_split, e_mfma))
        assert e_split <= 1.25 * e_mfma + 2e-8, (tag, name, e_split, e_mfma)
        check("%s split %s" % (tag, name), out[True][i], r, 3e-6)
    e_sums = ((out[True][3] - rs).abs().max() / rs.abs().max()).item()
    assert e_sums < 2e-7, (tag, "statistics", e_sums)
    assert not ops.streamk_error_codes()


@pytest.mark.parametrize("case", [("L2 3x3", 8, 40, 213, 64, 128, 3, 3, 1, 1), ("L1 4x4 s2 (256 x 64 tiles)", 8, 80, 427, 64, 64, 4, 4, 2, 1),
                                  ("L7 (6,3) valid", 32, 10, 53, 256, 256, 6, 3, 1, 0)], ids=lambda c: c[0])
def test_presplit_weights_are_bit_identical_to_the_in_kernel_split(ops, case):
    """Round 6: split-fp32 launches read the weights pre-split into three bf16 planes (made once per optimiser step with the transposed mirrors,
    sdt_wt_desc.planes = 3 -> sdt_convsk_f32_w3) instead of splitting them in every tile on every K step.  Same conversions, same exact differences:
    forward and input gradient must be BIT-identical to the launches that split the fp32 weights themselves, for a weight a mirror group has
    registered (as optim.FlatAdam does); and the planes must follow the weights when the optimiser steps (mark_dirty -> refresh)."""
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    assert ops.F32_SPLIT
    prev_w3, ops.W3_PRESPLIT = ops.W3_PRESPLIT, True  # (measured slower than splitting in the loader, profiles/r06_w3_ab.txt: not the default)
    try:
        _presplit_case(ops, case)
    finally:
        ops.W3_PRESPLIT = prev_w3


def _presplit_case(ops, case):
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Hi, Wi, Cin, generator=g).to(DEV)
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(DEV))
    y_plain = ops.conv_forward(x, w, None, s, p)  # not registered anywhere: the kernel splits
    gy = torch.randn(y_plain.shape, generator=g).to(DEV)
    dx_plain = ops.conv_input_grad(gy, w, x.shape, s, p)
    plan = ops._sk_plan(ops.conv_geom_for(x.shape, w, s, p), 1, 0, 1, x.device, forward=True)
    assert plan is not None and (plan.host[3] >> 26) & 1, "this shape is meant to take the split-fp32 kernel"
    group = ops.WeightMirrors([w])
    assert ops.WeightMirrors.lookup3(ops.weight_storage(w)) is not None and group.planes == 3
    y3 = ops.conv_forward(x, w, None, s, p)
    dx3 = ops.conv_input_grad(gy, w, x.shape, s, p)
    torch.cuda.synchronize()
    assert torch.equal(y3, y_plain) and torch.equal(dx3, dx_plain)
    # the three planes ARE the weight: hi + mid + lo == w exactly (fp32 sums of bf16 numbers in this order are exact)
    w3 = ops.WeightMirrors.lookup3(ops.weight_storage(w)).float()
    assert torch.equal((w3[0] + w3[1]) + w3[2], ops.weight_storage(w.detach()))
    # an optimiser step: the weights change behind the planes' back, mark_dirty is what FlatAdam.step calls
    with torch.no_grad():
        w.mul_(1.25)
    group.mark_dirty()
    y3b = ops.conv_forward(x, w, None, s, p)
    w_copy = torch.nn.Parameter(w.detach().clone())  # same values, registered nowhere: the kernel splits
    y_ref = ops.conv_forward(x, w_copy, None, s, p)
    torch.cuda.synchronize()
    assert torch.equal(y3b, y_ref) and not torch.equal(y3b, y_plain)
