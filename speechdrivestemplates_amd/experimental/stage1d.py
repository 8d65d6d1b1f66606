"""The Conv1d stage of the sdt generator (UNet_1D + decoder, reference generator.py:70-85,98-116) as ONE launch per layer and
direction (csrc/conv1d.hip): 17 forward launches and 24 backward launches on the step's critical path instead of ~41 + ~60.

Every layer stores its RAW conv output plus per-row partial statistics; the per-(b,t) normalisation over channels, the LeakyReLU and
the linear upsample + skip add of the decoder inputs are applied by the CONSUMING launch while it stages its A operand, and in
backward the normalisation backward is formed on load the same way.  Weight gradients are not on the critical path: the tensors
the generic weight-gradient kernels need (activated layer inputs, normalisation-backward outputs) are materialised on the side
stream, where they overlap the audio encoder's backward pass.

Used for the InstanceNorm ('IN') generator in exact-fp32 math while gradients are enabled; every other case (BatchNorm generator,
no-grad inference, the bf16 modes) keeps the per-block path of core.networks.building_blocks.
"""
import os

import torch

from .. import _lib, ops
from .._lib import check
from . import require

# Opt-in (SDT_STAGE1D=1): measured on MI355X (profiles/r02_conv1d_stage.txt) the fused stage shortens the exposed Conv1d chains
# (0.85 -> 0.81 ms per step) but moves more work onto the weight-gradient side stream, and the whole step comes out 2 % SLOWER
# (3950 vs 4040 clips/s), so the per-block path stays the default.
ENABLED = False  # switched by enable() (bench.py --fused-conv1d, the tests); no environment variable
_p = ops._p


class _Layer:
    __slots__ = ("name", "w", "bias", "k", "stride", "pad", "mode", "src", "src2", "Ti", "T2", "To", "cin", "cout", "norm")


def plan(gen, T):
    """Static launch plan of the stage for sequence length T: [(name, weight, kernel, stride, input mode, sources...)]."""
    layers = []

    def add(name, conv, mode, src, src2, Ti, T2, norm=True):
        L = _Layer()
        L.name, L.w, L.bias = name, conv.weight, getattr(conv, "bias", None)
        L.k, L.stride, L.pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        L.mode, L.src, L.src2, L.Ti, L.T2 = mode, src, src2, Ti, T2
        L.To = ops.out_size(Ti, L.k, L.stride, L.pad)
        L.cin, L.cout, L.norm = conv.weight.shape[1], conv.weight.shape[0], norm
        layers.append(L)
        return L.To

    u = gen.unet
    t = add("e0", u.e0.conv, 0, "h0", None, T, 0)
    t = add("e1", u.e1.conv, 1, "e0", None, t, 0)
    Ts = {"e1": t}
    for i in range(2, 7):
        t = add("e%d" % i, getattr(u, "e%d" % i).conv, 1, "e%d" % (i - 1), None, t, 0)
        Ts["e%d" % i] = t
    prev, tprev = "e6", t
    for i in (5, 4, 3, 2, 1):
        skip = "e%d" % i
        tprev_out = add("d%d" % i, getattr(u, "d%d" % i).conv, 2, skip, prev, Ts[skip], tprev)
        prev, tprev = "d%d" % i, tprev_out
    for j in range(4):
        tprev = add("dec%d" % j, gen.decoder[j].conv, 1, prev, None, tprev, 0)
        prev = "dec%d" % j
    add("head", gen.decoder[4], 1, prev, None, tprev, 0, norm=False)
    return layers


def usable(gen, h0):
    """The fused stage applies to the 'IN' generator in fp32 math with gradients enabled and kernel-layout weights."""
    if not (ENABLED and torch.is_grad_enabled() and h0.is_cuda and ops._CONV_MATH_NOW[0] == 0):
        return False
    if gen.unet.e0.norm_type != 'IN' or gen.decoder[0].norm_type != 'IN':
        return False
    convs = [getattr(gen.unet, n).conv for n in ("e0", "e1", "e2", "e3", "e4", "e5", "e6", "d5", "d4", "d3", "d2", "d1")]
    convs += [gen.decoder[j].conv for j in range(4)] + [gen.decoder[4]]
    for c in convs:
        w = c.weight
        if ops.weight_storage(w).data_ptr() != w.data_ptr():
            return False
        if w.shape[1] % 4 or (c is not gen.decoder[4] and w.shape[0] % 64):
            return False
    return h0.shape[1] >= 64 and h0.shape[1] % 32 == 0  # five stride-2 levels down to T/32 >= 2


# K slices inside the launch for layers with few output tiles (sdt_c1d.splitk: partial tiles in slabs, the last slice to arrive
# adds them in order).  Correct and deterministic, but MEASURED SLOWER and therefore off: the layers it targets went from 20 to
# 36-60 us per launch and the whole step lost 8 % -- on a multi-XCD part the device-scope release / acquire around the arrival
# counter is an L2 write-back + invalidate of the whole XCD (the slices of a tile run on different XCDs), which also hurts the
# kernels running next to it on the side stream (profiles/r02_conv1d_stage.txt).
SPLITK = False


def enable(on=True):
    """Route the generator's Conv1d stage through c1d_kernel (needs the tuning library)."""
    global ENABLED
    import sys
    if on:
        require("the fused Conv1d stage")
    ENABLED = bool(on)
    ops.STAGE1D = sys.modules[__name__] if on else None
_WS = {}        # device index -> (slab workspace, arrival counters): launches of one stream use it one after the other


def _split(d, M, dev):
    """K slices for a launch: spread the 12-16 serial K steps of a layer with few output tiles over the chip."""
    tiles = -(-M // 32) * -(-d.Cout // 64)
    steps = -(-(d.taps * d.Cin) // 64)
    s = 1 if (not SPLITK or tiles >= 192) else max(1, min(8, steps // 2, 256 // tiles))
    d.splitk = s
    if s > 1:
        need = s * M * d.Cout
        ws = _WS.get(dev.index)
        if ws is None or ws[0].numel() < need:
            cnt = ws[1] if ws is not None else torch.zeros(4096, device=dev, dtype=torch.int32)
            ws = (torch.empty(max(need, 1 << 22), device=dev, dtype=torch.float32), cnt)
            _WS[dev.index] = ws
        d.slabs, d.counters = _p(ws[0]), _p(ws[1])


def _launch(lib, role, d, st, M):
    """one c1d launch (+ a ConvProfiler record when bench.py samples this step)"""
    _split(d, M, torch.device("cuda", torch.cuda.current_device()))
    if ops.PROFILER is None:
        check(lib.sdt_c1d_layer_f32(d, st))
        return
    e0, e1 = ops.PROFILER.event(), ops.PROFILER.event()
    e0.record()
    check(lib.sdt_c1d_layer_f32(d, st))
    e1.record()
    flops = 2.0 * M * d.Cout * d.taps * d.Cin
    nbytes = 4.0 * (d.B * d.Ti * d.Cin + M * d.Cout + d.Cout * d.taps * d.Cin)
    ops.PROFILER.records.append(("c1d_kernel", role, False, flops, nbytes, e0, e1))


class Gen1dStageFn(torch.autograd.Function):
    """h0 (B,T,Cin0) -> prediction (B,T,2K): UNet_1D.forward + the decoder of SequenceGeneratorCNN (generator.py:70-85,113-116)."""

    @staticmethod
    def forward(ctx, h0, gen, *weights):
        lib = _lib.load()
        st = ops._stream()
        h0 = h0.contiguous()
        B, T = h0.shape[0], h0.shape[1]
        layers = plan(gen, T)
        dev = h0.device
        ys, stats = {"h0": h0}, {}
        slope, eps = float(gen.unet.e0.slope), ops.BN_EPS
        for L in layers:
            M = B * L.To
            y = torch.empty((B, L.To, L.cout), device=dev, dtype=torch.float32)
            np_out = L.cout // 64 if L.norm else 0
            s_out = torch.empty((M, np_out, 2), device=dev, dtype=torch.float32) if L.norm else None
            d = _lib.C1d()
            d.X, d.W, d.Y = _p(ys[L.src]), _p(L.w), _p(y)
            d.bias = _p(L.bias)
            d.ystats = _p(s_out)
            d.B, d.Ti, d.T2, d.Cin, d.To, d.Cout = B, L.Ti, L.T2, L.cin, L.To, L.cout
            d.taps, d.stride, d.pad, d.in_mode = L.k, L.stride, L.pad, L.mode
            d.eps, d.slope = eps, slope
            if L.mode >= 1:
                d.xstats, d.np_in = _p(stats[L.src]), stats[L.src].shape[1]
            if L.mode == 2:
                d.X2, d.x2stats, d.np_in2 = _p(ys[L.src2]), _p(stats[L.src2]), stats[L.src2].shape[1]
            _launch(lib, "fwd", d, st, M)
            ys[L.name] = y
            if L.norm:
                stats[L.name] = s_out
        ctx.gen, ctx.layers, ctx.ys, ctx.stats, ctx.dims = gen, layers, ys, stats, (B, T, slope, eps)
        return ys["head"]

    @staticmethod
    def backward(ctx, dpred):
        lib = _lib.load()
        st = ops._stream()
        gen, layers, ys, stats = ctx.gen, ctx.layers, ctx.ys, ctx.stats
        B, T, slope, eps = ctx.dims
        dev = dpred.device
        dpred = dpred.contiguous()
        by = {L.name: L for L in layers}
        dz, bst, skipg = {}, {}, {}   # gradient w.r.t. a layer's ACTIVATED output; its backward partials; skip-path gradients

        def new_bstats(name):
            return torch.empty((B * by[name].To, by[name].cout // 64, 2), device=dev, dtype=torch.float32)

        # ---- head: k1 conv with 2K = 242 input channels of the gradient (not a multiple of 4): generic kernel, then statistics
        head = by["head"]
        g = ops.conv_input_grad(dpred, head.w, (B, head.Ti, head.cin), 1, 0)
        src = head.src
        dz[src], bst[src] = torch.empty_like(g), new_bstats(src)
        check(lib.sdt_c1d_upsample_bwd_stats_f32(_p(g), _p(dz[src]), _p(ys[src]), _p(stats[src]), stats[src].shape[1], _p(bst[src]),
                                                 bst[src].shape[1], B, head.Ti, head.Ti, head.cin, eps, slope, st))
        # ---- the chain, last layer first
        for L in reversed(layers[:-1]):
            wt = ops.WeightMirrors.lookup(L.w)
            if wt is None:
                wt = torch.empty((L.cin, L.k, L.cout), device=dev, dtype=torch.float32)
                check(lib.sdt_weight_transpose_f32(_p(L.w), _p(wt), L.cout, L.k, L.cin, st))
            M = B * L.Ti
            out = torch.empty((B, L.Ti, L.cin), device=dev, dtype=torch.float32)
            d = _lib.C1d()
            d.X, d.X2, d.xstats, d.x2stats = _p(dz[L.name]), _p(ys[L.name]), _p(stats[L.name]), _p(bst[L.name])
            d.np_in, d.np_in2 = stats[L.name].shape[1], bst[L.name].shape[1]
            d.W, d.Y = _p(wt), _p(out)
            d.B, d.Ti, d.T2, d.Cin, d.To, d.Cout = B, L.To, 0, L.cout, L.Ti, L.cin
            d.taps, d.stride, d.pad, d.in_mode = L.k, L.stride, L.pad, 3
            d.eps, d.slope = eps, slope
            if L.mode == 1:  # the layer below is normalised: total gradient of its output (+ skip path) and its backward partials
                below = L.src
                if below in skipg:
                    d.add = _p(skipg[below])
                bst[below] = new_bstats(below)
                d.ystats, d.bw_y, d.bw_stats, d.np_bw = _p(bst[below]), _p(ys[below]), _p(stats[below]), stats[below].shape[1]
                _launch(lib, "dX", d, st, M)
                dz[below] = out
            elif L.mode == 2:  # decoder input = upsample(prev) + skip: `out` is the gradient of the sum
                _launch(lib, "dX", d, st, M)
                skipg[L.src] = out
                prev = L.src2
                dz[prev], bst[prev] = torch.empty((B, L.T2, L.cin), device=dev, dtype=torch.float32), new_bstats(prev)
                check(lib.sdt_c1d_upsample_bwd_stats_f32(_p(out), _p(dz[prev]), _p(ys[prev]), _p(stats[prev]), stats[prev].shape[1],
                                                         _p(bst[prev]), bst[prev].shape[1], B, L.T2, L.Ti, L.cin, eps, slope, st))
            else:  # first layer: plain input, its gradient leaves the stage
                _launch(lib, "dX", d, st, M)
                dh0 = out
        # ---- weight gradients: off the critical path, on the side stream (they overlap the audio encoder's backward pass)
        side = ops._side_stream() if (ops.OVERLAP_DW and ops._side_ok()) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        ctxm = torch.cuda.stream(side) if side is not None else _Null()
        keep = []
        with ctxm:
            ss = ops._stream()
            ops.stage_mark("dw1d:begin")
            z, mean, rstd = {"h0": ys["h0"]}, {}, {}
            for L in layers[:-1]:
                y = ys[L.name]
                z[L.name], mean[L.name], rstd[L.name] = torch.empty_like(y), torch.empty(B * L.To, device=dev), torch.empty(B * L.To, device=dev)
                check(lib.sdt_c1d_rownorm_partials_f32(_p(y), _p(stats[L.name]), stats[L.name].shape[1], _p(z[L.name]), _p(mean[L.name]),
                                                       _p(rstd[L.name]), B * L.To, L.cout, eps, slope, ss))
            for L in layers:
                if L.mode == 2:
                    xin = torch.empty((B, L.Ti, L.cin), device=dev, dtype=torch.float32)
                    check(lib.sdt_upsample_add_fwd_f32(_p(z[L.src2]), _p(z[L.src]), _p(xin), B, L.T2, L.Ti, L.cin, ss))
                else:
                    xin = z[L.src]
                if L.norm:
                    gy = torch.empty_like(ys[L.name])
                    check(lib.sdt_rownorm_bwd_f32(_p(dz[L.name]), _p(ys[L.name]), _p(mean[L.name]), _p(rstd[L.name]), _p(gy), B * L.To, L.cout,
                                                  slope, ss))
                else:
                    gy = dpred
                    if L.bias is not None and L.bias.requires_grad:
                        check(lib.sdt_col_sum_f32(_p(gy), _p(ops.grad_buffer(L.bias)), B * L.To, L.cout, ss))
                if L.w.requires_grad:
                    ops.conv_weight_grad(xin, gy, L.w, L.stride, L.pad)
                keep += [xin, gy]
            ops.stage_mark("dw1d:end")
        if side is not None:
            for t in keep + list(dz.values()) + list(z.values()) + list(mean.values()) + list(rstd.values()) + [dpred]:
                t.record_stream(side)
            for t in list(ys.values()) + list(stats.values()):
                t.record_stream(side)
        return (dh0, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
