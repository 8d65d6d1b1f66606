"""CPU tests of the host logic: the tap-table geometries handed to sdt_conv_taps_f32 / sdt_conv_dw_f32
reproduce nn.Conv{1,2}d forward, input-gradient and weight-gradient when interpreted literally (a torch
emulator of the kernel contract written out in include/sdt_hip.h), and the C-ABI library exports every
symbol the header declares.  No compute call is made into the library (no GPU here)."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

from conftest import REPO
from speechdrivestemplates_amd import _lib, ops


def emulate_conv_taps(x, w_store, g, y):
    """Literal statement of the sdt_conv_taps_f32 contract on CPU tensors (x (B,Hi,Wi,Cin), w (Cout,Tw,Cin))."""
    oy = torch.arange(g.Ho)
    ox = torch.arange(g.Wo)
    acc = torch.zeros(g.B, g.Ho, g.Wo, g.Cout, dtype=x.dtype)
    for t in range(g.ntaps):
        iy, ix = oy * g.sy + g.dy[t], ox * g.sx + g.dx[t]
        vy, vx = (iy >= 0) & (iy < g.Hi), (ix >= 0) & (ix < g.Wi)
        if not vy.any() or not vx.any():
            continue
        xs = x[:, iy[vy]][:, :, ix[vx]]
        contrib = torch.einsum("bhwc,nc->bhwn", xs, w_store[:, g.wt[t], :])
        sub = acc[:, vy]
        sub[:, :, vx] += contrib
        acc[:, vy] = sub
    y[:, g.ooy::g.osy, g.oox::g.osx][:, :g.Ho, :g.Wo] = acc
    return y


def emulate_conv_dw(x, dy_full, g, dw):
    oy = torch.arange(g.Ho)
    ox = torch.arange(g.Wo)
    dysel = dy_full[:, g.ooy::g.osy, g.oox::g.osx][:, :g.Ho, :g.Wo]
    for t in range(g.ntaps):
        iy, ix = oy * g.sy + g.dy[t], ox * g.sx + g.dx[t]
        vy, vx = (iy >= 0) & (iy < g.Hi), (ix >= 0) & (ix < g.Wi)
        if not vy.any() or not vx.any():
            continue
        xs = x[:, iy[vy]][:, :, ix[vx]]
        ds = dysel[:, vy][:, :, vx]
        dw[:, g.wt[t], :] += torch.einsum("bhwn,bhwc->nc", ds, xs)
    return dw


CASES_2D = [  # (Hi, Wi, Cin, Cout, kh, kw, s, p) -- the three reference shapes (building_blocks.py:8-12, generator.py:29)
    (10, 13, 3, 5, 3, 3, 1, 1), (10, 13, 4, 6, 4, 4, 2, 1), (9, 11, 4, 6, 4, 4, 2, 1), (7, 9, 2, 3, 6, 3, 1, 0),
    (10, 53, 2, 2, 6, 3, 1, 0), (8, 8, 1, 4, 3, 3, 1, 1),
]


@pytest.mark.parametrize("case", CASES_2D)
def test_conv2d_geometries(case):
    Hi, Wi, Cin, Cout, kh, kw, s, p = case
    torch.manual_seed(0)
    B = 2
    x = torch.randn(B, Cin, Hi, Wi, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, kh, kw, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, s, p)
    gy = torch.randn_like(y)
    y.backward(gy)
    x_cl, gy_cl = ops.cl(x.detach()), ops.cl(gy)
    ws = ops.weight_storage(w.detach())
    g = ops.fwd_geom(B, Hi, Wi, Cin, Cout, kh, kw, s, p)
    y_em = emulate_conv_taps(x_cl, ws, g, torch.full((B, g.Hy, g.Wy, Cout), float("nan"), dtype=torch.float64))
    assert torch.allclose(ops.cf_view(y_em), y.detach(), atol=1e-10)
    # input gradient: parity-class tap-convs of dY with the transposed weights cover every dX element once
    wt = ws.permute(2, 1, 0).contiguous()
    dx_em = torch.full((B, Hi, Wi, Cin), float("nan"), dtype=torch.float64)
    for gg, (py, px) in ops.dx_geoms(B, Hi, Wi, Cin, Cout, kh, kw, s, p):
        if gg is None:
            dx_em[:, py::s, px::s] = 0
        else:
            emulate_conv_taps(gy_cl, wt, gg, dx_em)
    assert not torch.isnan(dx_em).any()
    assert torch.allclose(ops.cf_view(dx_em), x.grad, atol=1e-10)
    dw_em = emulate_conv_dw(x_cl, gy_cl, g, torch.zeros(Cout, kh * kw, Cin, dtype=torch.float64))
    assert torch.allclose(dw_em, ops.weight_storage(w.grad), atol=1e-9)


@pytest.mark.parametrize("case", [(16, 6, 5, 3, 1, 1), (16, 6, 5, 4, 2, 1), (15, 6, 5, 4, 2, 1), (2, 4, 4, 4, 2, 1), (9, 3, 2, 1, 1, 0)])
def test_conv1d_geometries(case):
    T, Cin, Cout, k, s, p = case
    torch.manual_seed(1)
    B = 3
    x = torch.randn(B, Cin, T, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(x, w, None, s, p)
    gy = torch.randn_like(y)
    y.backward(gy)
    x4, gy4 = ops.cl(x.detach()).unsqueeze(1), ops.cl(gy).unsqueeze(1)
    ws = ops.weight_storage(w.detach())
    g = ops.conv_geom_for(x4.shape, w, s, p)
    y_em = emulate_conv_taps(x4, ws, g, torch.zeros(B, 1, g.Wy, Cout, dtype=torch.float64))
    assert torch.allclose(y_em.squeeze(1).permute(0, 2, 1), y.detach(), atol=1e-10)
    wt = ws.permute(2, 1, 0).contiguous()
    dx_em = torch.full((B, 1, T, Cin), float("nan"), dtype=torch.float64)
    for gg, (py, px) in ops.dx_geoms_1d(B, T, Cin, Cout, k, s, p):
        if gg is None:
            dx_em[:, :, px::s] = 0
        else:
            emulate_conv_taps(gy4, wt, gg, dx_em)
    assert torch.allclose(dx_em.squeeze(1).permute(0, 2, 1), x.grad, atol=1e-10)
    dw_em = emulate_conv_dw(x4, gy4, g, torch.zeros(Cout, k, Cin, dtype=torch.float64))
    assert torch.allclose(dw_em, ops.weight_storage(w.grad), atol=1e-9)


def test_stft_as_tap_conv_matches_torch_stft():
    """The mel front end runs the STFT as a 3-tap conv over 160-sample hops with a windowed DFT basis."""
    from oracle import sdt_oracle as O
    torch.manual_seed(0)
    L = 160 * 20 + 106
    a = 0.1 * torch.randn(2, L, dtype=torch.float64)
    F_ = 1 + L // 160
    nh = F_ + 2
    pad = F.pad(a.unsqueeze(1), (256, 256), mode="reflect").squeeze(1)
    hops = torch.zeros(2, nh * 160, dtype=torch.float64)
    n = min(nh * 160, pad.shape[1] - 56)
    hops[:, :n] = pad[:, 56:56 + n]
    basis = ops.dft_basis(O.mel_window(torch.float64)).double()
    g = ops._geom(B=2, Hi=1, Wi=nh, Cin=160, Ho=1, Wo=F_, Hy=1, Wy=F_, Cout=514, sy=1, sx=1, osy=1, osx=1, ooy=0, oox=0, Tw=3,
                  taps=[(0, 0, 0), (0, 1, 1), (0, 2, 2)])
    spec = emulate_conv_taps(hops.reshape(2, 1, nh, 160), basis, g, torch.zeros(2, 1, F_, 514, dtype=torch.float64)).squeeze(1)
    power = (spec[..., 0::2] ** 2 + spec[..., 1::2] ** 2).transpose(1, 2)
    ref = O.stft_power(a, O.mel_window(torch.float64))
    assert power.shape == ref.shape
    assert (power - ref).abs().max() / ref.abs().max() < 1e-6  # basis is stored in fp32


def test_weight_layout_roundtrip():
    w = torch.randn(6, 4, 3, 5)
    wl = ops.to_weight_layout(w)
    assert torch.equal(wl, w) and wl.permute(0, 2, 3, 1).is_contiguous()
    assert ops.weight_storage(wl).data_ptr() == wl.data_ptr()
    assert torch.zeros_like(wl).stride() == wl.stride()
    w1 = ops.to_weight_layout(torch.randn(6, 4, 3))
    assert ops.weight_storage(w1).data_ptr() == w1.data_ptr() and ops.weight_storage(w1).shape == (6, 3, 4)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "sdt_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(sdt_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libsdt_hip.so does not export %s" % name
    extra = {"sdt_last_error", "sdt_abi_version", "sdt_get_conv_math", "sdt_conv_dw_workspace_bytes", "sdt_convsk_plan_bytes",
             "sdt_convsk_workspace_bytes", "sdt_convsk_dw_plan_bytes", "sdt_convsk_dw_workspace_bytes"}
    declared |= set(re.findall(r"^\s*int64_t\s+(sdt_\w+)\s*\(", hdr, flags=re.M))
    for name in sorted(declared):
        assert hasattr(lib, name), "libsdt_hip.so does not export %s" % name
    assert declared - extra == set(_lib.SIGNATURES), "ctypes signatures out of sync with the header: %s" % ((declared - extra) ^ set(_lib.SIGNATURES))
    assert _lib.load().sdt_abi_version() == 1
    # the experiments' header is matched by the tuning library (when it has been built) and by nothing in the product library
    ehdr = open(os.path.join(REPO, "include", "sdt_hip_experimental.h")).read()
    edecl = set(re.findall(r"^\s*int\s+(sdt_\w+)\s*\(", ehdr, flags=re.M))
    assert edecl == set(_lib.EXPERIMENTAL_SIGNATURES), edecl ^ set(_lib.EXPERIMENTAL_SIGNATURES)
    product = ctypes.CDLL(os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip.so"))
    for name in sorted(edecl):
        assert not hasattr(product, name), "the product library exports the experiment %s" % name
    tuning = os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip_tuning.so")
    if os.path.exists(tuning):
        tl = ctypes.CDLL(tuning)
        for name in sorted(edecl | declared):
            assert hasattr(tl, name), "libsdt_hip_tuning.so does not export %s" % name


def test_ops_refuse_cpu_tensors():
    x = torch.zeros(1, 4, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.RowNormActFn.apply(x, 0.2)


def test_fgd_matches_closed_form():
    """FGD of two Gaussians with diagonal covariances has a closed form; also symmetric and zero on identical sets."""
    import numpy as np
    from speechdrivestemplates_amd.fgd import compute_fgd
    rng = np.random.default_rng(0)
    n, d = 200000, 6
    s1, s2 = np.array([1.0, 0.5, 2.0, 1.5, 0.7, 1.1]), np.array([0.8, 0.9, 1.0, 2.0, 0.6, 1.4])
    m = np.array([0.3, -0.2, 0.0, 0.5, 0.1, -0.4])
    a, b = rng.standard_normal((n, d)) * s1, rng.standard_normal((n, d)) * s2 + m
    exact = (m ** 2).sum() + ((s1 - s2) ** 2).sum()
    got = compute_fgd(a, b)
    assert abs(got - exact) < 0.03 * exact, (got, exact)
    assert abs(compute_fgd(a, b) - compute_fgd(b, a)) < 1e-9 and abs(compute_fgd(a, a)) < 1e-8


def test_fgd_matches_reference_fixture():
    """compute_fgd against outputs of the reference's own core/utils/fgd.py:59-64 (tests/golden/make_golden.py section 5):
    full-rank, correlated and rank-deficient (n < d: singular covariance product) code sets."""
    import os

    import numpy as np

    from conftest import GOLDEN
    from speechdrivestemplates_amd.fgd import compute_fgd
    g = dict(np.load(os.path.join(GOLDEN, "fgd.npz")))
    tags = sorted({k.split("/")[0] for k in g})
    assert len(tags) == 4
    for tag in tags:
        ref = float(g[tag + "/fgd_ab"][0])  # the reference returns a float32 tensor
        got = compute_fgd(g[tag + "/a"], g[tag + "/b"])
        assert abs(got - ref) <= 2e-6 * abs(ref) + 1e-6, (tag, got, ref)
        if tag + "/fgd_aa" in g:
            assert abs(compute_fgd(g[tag + "/a"], g[tag + "/a"]) - float(g[tag + "/fgd_aa"][0])) <= 1e-5, tag
