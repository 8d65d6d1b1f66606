"""CPU tests of the host logic: the tap-table geometries handed to sdt_conv_taps_f32 / sdt_conv_dw_f32
reproduce nn.Conv{1,2}d forward, input-gradient and weight-gradient when interpreted literally (a torch
emulator of the kernel contract written out in include/sdt_hip.h), and the C-ABI library exports every
symbol the header declares.  No compute call is made into the library (no GPU here)."""
import ctypes
import os
import re

import numpy as np
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
             "sdt_convsk_workspace_bytes", "sdt_convsk_dw_plan_bytes", "sdt_convsk_dw_workspace_bytes", "sdt_convsk_plan_bytes_t",
             "sdt_convsk_dw_plan_bytes_t", "sdt_convsk_get_spin_limit", "sdt_conv_dw_group_plan_bytes"}
    declared |= set(re.findall(r"^\s*(?:int64_t|unsigned)\s+(sdt_\w+)\s*\(", hdr, flags=re.M))
    for name in sorted(declared):
        assert hasattr(lib, name), "libsdt_hip.so does not export %s" % name
    assert declared - extra == set(_lib.SIGNATURES), "ctypes signatures out of sync with the header: %s" % ((declared - extra) ^ set(_lib.SIGNATURES))
    assert _lib.load().sdt_abi_version() == _lib.ABI_VERSION == 5
    # the debug / fault-injection hooks live in the -DSDT_TUNING library only
    product = ctypes.CDLL(os.path.join(REPO, "speechdrivestemplates_amd", "lib", "libsdt_hip.so"))
    for name in ("sdt_debug_convsk_mute_range", "sdt_debug_set_timeline_sk", "sdt_debug_spin"):
        assert not hasattr(product, name), "the product library exports the debug hook %s" % name


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


def _plan_tables(ops, garr, n, rpg, bwd_groups, reserve=0):
    """Build a stream-K plan on the host (no GPU involved) and return its header and tables as numpy arrays."""
    lib = _lib.load()
    assert lib.sdt_convsk_supported(garr, n) == 1
    assert lib.sdt_convsk_set_reserved_slots(reserve) == 0
    try:
        nbytes = lib.sdt_convsk_plan_bytes(garr, n)
        blob = (ctypes.c_int32 * (nbytes // 4))()
        assert lib.sdt_convsk_plan_build(garr, n, rpg, bwd_groups, ctypes.addressof(blob), nbytes) == 0, lib.sdt_last_error()
    finally:
        lib.sdt_convsk_set_reserved_slots(0)
    P = np.frombuffer(blob, dtype=np.int32).copy()
    hdr = dict(bm=int(P[1]), bn=int(P[2]), G=int(P[3] & 0xffff), wpc=int(P[3] >> 16), ncls=int(P[4]), nnb=int(P[5] & 0xffff), ntmajor=int(P[5] >> 16),
               T=int(P[6]), S=int(P[7]), rows=int(P[8]), mts=int(P[9]))
    rowinfo = P[P[10]:P[10] + 4 * hdr["rows"]].reshape(-1, 4)
    tileinfo = P[P[11]:P[11] + 2 * hdr["mts"]].reshape(-1, 2)
    tilecum = P[P[12]:P[12] + hdr["T"] + 1]
    range_tile = P[P[13]:P[13] + hdr["G"]]
    return hdr, rowinfo, tileinfo, tilecum, range_tile


@pytest.mark.parametrize("case", [("3x3 pad 1, 10-row images (row-major, n-tile-major)", 32, 10, 53, 256, 256, 3, 3, 1, 1, "fwd", 0),
                                  ("(6,3) valid: input gradient (row-major, n-tile-major)", 32, 10, 53, 256, 256, 6, 3, 1, 0, "dx", 32),
                                  ("4x4 stride 2: input gradient, 4 parity classes", 8, 40, 213, 128, 128, 4, 4, 2, 1, "dx", 0),
                                  ("3x3 pad 1 with HALF of the GPU reserved (two processes on one GPU)", 32, 20, 106, 128, 256, 3, 3, 1, 1, "fwd", 256),
                                  ("3x3 pad 1, 40-row images (natural order)", 8, 40, 213, 64, 128, 3, 3, 1, 1, "fwd", 0)], ids=lambda c: c[0])
def test_streamk_plan_is_a_partition_of_the_work(case):
    """sdt_convsk_plan_build is host code: every output row appears exactly once with the offsets / tap masks a brute-force statement gives,
    whatever ORDER the plan chose for the rows; a tile's live-tap mask is the union over its rows; the step prefix sums, the split into G
    ranges (with reserved slots: fewer ranges) and the first tile of every range are consistent."""
    from speechdrivestemplates_amd import ops
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p, role, reserve = case
    if role == "fwd":
        g = ops.fwd_geom(B, Hi, Wi, Cin, Cout, kh, kw, s, p)
        garr, n, gs = g, 1, [g]
        rpg = g.Ho * g.Wo  # InstanceNorm groups: one per clip
    else:
        garr, n, gs = ops.dx_pack(B, Hi, Wi, Cin, Cout, kh, kw, s, p, False)
        rpg = -1
    hdr, rowinfo, tileinfo, tilecum, range_tile = _plan_tables(ops, garr, n, rpg, B if role == "dx" else 1, reserve)
    bm, nnb = hdr["bm"], hdr["nnb"]
    assert hdr["G"] == (512 - reserve if hdr["wpc"] == 2 else (256 - reserve // 2) & ~7) and hdr["ncls"] == n and hdr["bn"] * nnb == gs[0].Cout
    SK_OOB = -(1 << 31)
    row0 = mt0 = 0
    total_live = 0
    per_mt_live = []
    for g in gs:
        M = g.B * g.Ho * g.Wo
        nmb = -(-M // bm)
        ri = rowinfo[row0:row0 + nmb * bm]
        real = ri[ri[:, 2] != SK_OOB]
        assert len(real) == M and (ri[:, 2] == SK_OOB).sum() == nmb * bm - M          # padding rows only past the last real row
        # brute force per output position (b, oy, ox), keyed by the byte offset of its output row
        b, oy, ox = np.meshgrid(np.arange(g.B), np.arange(g.Ho), np.arange(g.Wo), indexing="ij")
        yoff = (((b * g.Hy + oy * g.osy + g.ooy) * g.Wy + ox * g.osx + g.oox) * g.Cout * 4).ravel().astype(np.int64)
        xoff = (((b * g.Hi + oy * g.sy) * g.Wi + ox * g.sx) * g.Cin * 4).ravel().astype(np.int64)
        inval = np.full(M, 1 << 31, dtype=np.int64)
        for t in range(g.ntaps):
            iy, ix = oy * g.sy + g.dy[t], ox * g.sx + g.dx[t]
            bad = ((iy < 0) | (iy >= g.Hi) | (ix < 0) | (ix >= g.Wi)).ravel()
            inval |= bad.astype(np.int64) << t
        grp = (b.ravel() if (role == "dx" or rpg == g.Ho * g.Wo) else np.zeros(M, dtype=np.int64))
        order = np.argsort(yoff)
        got = real[np.argsort(real[:, 2].astype(np.int64))]
        assert np.array_equal(got[:, 2].astype(np.int64), yoff[order]), "every output row exactly once"
        assert np.array_equal(got[:, 0].astype(np.int64) & 0xffffffff, xoff[order] & 0xffffffff)
        assert np.array_equal(got[:, 1].astype(np.int64) & 0xffffffff, inval[order] & 0xffffffff)
        assert np.array_equal(got[:, 3].astype(np.int64), grp[order])
        # which order did the plan choose?  row-major: the rows after the first image row of clip 0 are the same image row of clip 1
        if g.B > 1 and g.Ho > 1:
            second_run = int(ri[g.Wo, 2])
            row_major = second_run == int(yoff.reshape(g.B, g.Ho, g.Wo)[1, 0, 0])
            assert row_major == ("row-major" in tag), (tag, "row-major" if row_major else "natural")
        # a group's rows come in runs of >= 32 consecutive plan rows, at most two groups per 32-row block (what the statistics epilogues assume)
        blocks = ri[:, 3].reshape(-1, 32)
        assert all(len(set(int(v) for v in blk if v >= 0)) <= 2 for blk in blocks)
        # tiles: the live mask (un-rotated) is the union of the valid taps of the tile's rows; culling is skipped on short tap lists
        nkc = g.Cin // 32
        for mt in range(nmb):
            rows = ri[mt * bm:(mt + 1) * bm]
            rows = rows[rows[:, 2] != SK_OOB]
            union = 0
            for v in rows[:, 1].astype(np.int64) & ((1 << g.ntaps) - 1):
                union |= ((1 << g.ntaps) - 1) & ~int(v)
            if g.ntaps <= 4:
                union = (1 << g.ntaps) - 1
            rmask, rot = int(tileinfo[mt0 + mt, 0]) & 0xffffffff, int(tileinfo[mt0 + mt, 1])
            unrot = 0
            for i in range(g.ntaps):
                if rmask >> i & 1:
                    unrot |= 1 << ((rot + i) % g.ntaps)
            assert unrot == union, (tag, mt, bin(unrot), bin(union))
            per_mt_live.append(bin(union).count("1") * nkc)
        total_live += sum(per_mt_live[mt0:mt0 + nmb]) * nnb
        row0 += nmb * bm
        mt0 += nmb
    assert hdr["S"] == total_live == int(tilecum[-1]) and tilecum[0] == 0 and np.all(np.diff(tilecum) > 0)
    # tile -> m-tile in the plan's tile order; per-tile step counts match the m-tile's live steps
    steps = np.diff(tilecum)
    assert bool(hdr["ntmajor"]) == ("n-tile-major" in tag), (tag, hdr["ntmajor"])
    if hdr["ntmajor"]:
        assert n == 1
        expect = np.tile(np.array(per_mt_live), nnb)
    else:
        expect = np.repeat(np.array(per_mt_live), nnb)
    assert np.array_equal(steps, expect)
    # ranges: range r starts at step floor(r * S / G); its first tile is the one that contains that step
    G, S = hdr["G"], hdr["S"]
    for r in range(G):
        s0 = r * S // G
        t = int(range_tile[r])
        assert tilecum[t] <= s0 < tilecum[t + 1], (r, s0, t)


@pytest.mark.parametrize("case", [("128x128 tiles", 32, 20, 106, 128, 256, 3, 3, 1, 1, 128, 128), ("64-row tiles (Cout 64)", 32, 80, 427, 64, 64, 4, 4, 2, 1, 64, 128),
                                  ("64-column tiles (9 x 64 columns)", 32, 40, 213, 64, 128, 3, 3, 1, 1, 128, 64)], ids=lambda c: c[0])
def test_streamk_weight_gradient_plan(case):
    """sdt_convsk_dw_plan_build (host code): tile shape by divisibility, one row-table entry per output position (offset of its (0,0) tap in X,
    invalid-tap mask, offset of its dY row), K steps of 32 positions, at least 8 steps per (tile, K-chunk) unit."""
    from speechdrivestemplates_amd import ops
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p, bm, bn = case
    g = ops.fwd_geom(B, Hi, Wi, Cin, Cout, kh, kw, s, p)
    lib = _lib.load()
    assert lib.sdt_convsk_dw_supported(g) == 1
    nbytes = lib.sdt_convsk_dw_plan_bytes(g)
    blob = (ctypes.c_int32 * (nbytes // 4))()
    assert lib.sdt_convsk_dw_plan_build(g, ctypes.addressof(blob), nbytes) == 0, lib.sdt_last_error()
    P = np.frombuffer(blob, dtype=np.int32)
    M = g.B * g.Ho * g.Wo
    K, T, G = int(P[9]), int(P[6]), int(P[3] & 0xffff)
    assert (int(P[1]), int(P[2])) == (bm, bn) and K == -(-M // 32) and T == (Cout // bm) * (g.ntaps * Cin // bn) and G == 512
    assert T <= G and K >= 8 * (G // T)
    ri = P[P[10]:P[10] + 4 * int(P[8])].reshape(-1, 4)
    b, oy, ox = np.meshgrid(np.arange(g.B), np.arange(g.Ho), np.arange(g.Wo), indexing="ij")
    xoff = (((b * g.Hi + oy * g.sy) * g.Wi + ox * g.sx) * g.Cin * 4).ravel().astype(np.int64)
    inval = np.full(M, 1 << 31, dtype=np.int64)
    for t in range(g.ntaps):
        iy, ix = oy * g.sy + g.dy[t], ox * g.sx + g.dx[t]
        inval |= ((iy < 0) | (iy >= g.Hi) | (ix < 0) | (ix >= g.Wi)).ravel().astype(np.int64) << t
    assert np.array_equal(ri[:M, 0].astype(np.int64) & 0xffffffff, xoff & 0xffffffff)
    assert np.array_equal(ri[:M, 1].astype(np.int64) & 0xffffffff, inval & 0xffffffff)
    assert np.array_equal(ri[:M, 2].astype(np.int64), np.arange(M, dtype=np.int64) * g.Cout * 4)
    assert np.all(ri[M:, 2] == -(1 << 31)) and np.all(ri[M:, 1] == -1)  # rows past M: every tap invalid, dY offset out of range


def test_streamk_plan_rejects_tiles_without_live_steps():
    """a k5 / stride-2 input gradient over a size that leaves trailing input rows no output position reaches puts whole tiles of a
    parity class at ZERO live K steps (in the image-row-major order); a range that ends exactly at such a tile would never visit it and its
    outputs -- zeros -- would stay unwritten.  The plan builder refuses such packs (ops._sk_plan then routes the launch to the 64x64 kernel of
    conv.hip, which writes every position; tests/test_ops_gpu.py::test_input_gradient_with_unreachable_rows checks the result)."""
    from speechdrivestemplates_amd import ops
    lib = _lib.load()
    arr, n, gs = ops.dx_pack(32, 10, 65, 128, 128, 5, 5, 2, 0, False)
    assert lib.sdt_convsk_supported(arr, n) == 1
    nbytes = lib.sdt_convsk_plan_bytes(arr, n)
    blob = (ctypes.c_int32 * (nbytes // 4))()
    assert lib.sdt_convsk_plan_build(arr, n, -1, 1, ctypes.addressof(blob), nbytes) != 0
    assert b"no live K step" in lib.sdt_last_error()
    import torch
    assert ops._sk_plan(arr, n, -1, 1, torch.device("cuda", 0)) is None  # the host-side cache records "no plan": no GPU is touched for that


@pytest.mark.parametrize("case", [("bf16 3x3 pad 1, 20-row images", 32, 20, 106, 128, 256, 3, 3, 1, 1, "fwd"),
                                  ("bf16 4x4 stride 2: input gradient, 4 parity classes", 8, 40, 213, 128, 128, 4, 4, 2, 1, "dx")], ids=lambda c: c[0])
def test_streamk_plan_in_bf16_units(case):
    """sdt_convsk_plan_build_t(.., SDT_BF16, SDT_BF16, ..): the same plan with every byte offset in 2-byte elements and a K step of 64 channels
    (128 bytes of an input row): row offsets, output offsets and the per-tile step counts against the brute-force statement."""
    from speechdrivestemplates_amd import ops
    lib = _lib.load()
    tag, B, Hi, Wi, Cin, Cout, kh, kw, s, p, role = case
    if role == "fwd":
        g = ops.fwd_geom(B, Hi, Wi, Cin, Cout, kh, kw, s, p)
        garr, n, gs, rpg = g, 1, [g], g.Ho * g.Wo
    else:
        garr, n, gs = ops.dx_pack(B, Hi, Wi, Cin, Cout, kh, kw, s, p, False)
        rpg = -1
    assert lib.sdt_convsk_supported_t(garr, n, _lib.BF16) == 1
    nbytes = lib.sdt_convsk_plan_bytes_t(garr, n, _lib.BF16)
    blob = (ctypes.c_int32 * (nbytes // 4))()
    assert lib.sdt_convsk_plan_build_t(garr, n, rpg, B if role == "dx" else 1, _lib.BF16, _lib.BF16, ctypes.addressof(blob), nbytes) == 0, lib.sdt_last_error()
    P = np.frombuffer(blob, dtype=np.int32).copy()
    assert (P[3] >> 24) & 3 == 3 and (P[3] >> 16) & 0xff == 2  # bf16 operands, bf16 output, two workgroups per CU
    bm, rows, T = int(P[1]), int(P[8]), int(P[6])
    rowinfo = P[P[10]:P[10] + 4 * rows].reshape(-1, 4)
    tilecum = P[P[12]:P[12] + T + 1]
    SK_OOB = -(1 << 31)
    row0 = 0
    for g in gs:
        M = g.B * g.Ho * g.Wo
        nmb = -(-M // bm)
        ri = rowinfo[row0:row0 + nmb * bm]
        real = ri[ri[:, 2] != SK_OOB]
        assert len(real) == M
        b, oy, ox = np.meshgrid(np.arange(g.B), np.arange(g.Ho), np.arange(g.Wo), indexing="ij")
        yoff = (((b * g.Hy + oy * g.osy + g.ooy) * g.Wy + ox * g.osx + g.oox) * g.Cout * 2).ravel().astype(np.int64)
        xoff = (((b * g.Hi + oy * g.sy) * g.Wi + ox * g.sx) * g.Cin * 2).ravel().astype(np.int64)
        order = np.argsort(yoff)
        got = real[np.argsort(real[:, 2].astype(np.int64))]
        assert np.array_equal(got[:, 2].astype(np.int64), yoff[order]) and np.array_equal(got[:, 0].astype(np.int64) & 0xffffffff, xoff[order] & 0xffffffff)
        row0 += nmb * bm
    assert np.all(np.diff(tilecum) > 0) and np.all(np.diff(tilecum) % (gs[0].Cin // 64) == 0)  # live taps x (Cin / 64) steps per tile
    cls = P[P[14]:P[14] + 11 + 3 * 20]
    assert cls[5] == gs[0].Cin // 64 and cls[11 + 1] == (gs[0].dy[1] * gs[0].Wi + gs[0].dx[1]) * gs[0].Cin * 2  # K steps per tap, tap shift in bytes
