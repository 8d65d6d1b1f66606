"""Hunt for the source of a non-finite value in the 2-ranks-on-1-GPU data-parallel harness (GPUTEST_r05: tests/test_dp_gpu.py
::test_two_ranks_one_gpu_bf16_storage_and_graph_replay[bf16-False] ended with all-NaN weights and clean error words on the driver's box).

  python tools/debug/dp_nan_hunt.py --iters 10 --storage bf16 [--graph] [--poison] [--trace] [--reserve 248] [--world 2]

Each iteration spawns ``world`` ranks over gloo that share cuda:0 and step the sdt_vae pipeline four times, as the test does.  After every step a
rank checks its losses, every optimiser's flat gradient and flat weights for non-finite values and reads every error word; the first offender is
reported with the names of the parameters it covers.
  --poison : torch.empty / empty_like / new_empty return NaN-filled memory (floating types) -- a kernel that reads a buffer it never wrote, or
             relies on a previous tenant's zeros, turns into a NaN deterministically instead of depending on what the box's memory held.
  --trace  : every autograd.Function of ops.py is wrapped: synchronise + check its outputs (forward) / gradients (backward); the first
             non-finite tensor is reported with the op's name.  (Serialises the streams: changes the timing.)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def _poison_allocations():
    import torch
    nan = float("nan")
    _empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def fill(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(nan)
            elif t.dtype in (torch.int32, torch.int64, torch.uint8, torch.int16):
                t.fill_(0x7f)
        return t

    torch.empty = lambda *a, **k: fill(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: fill(_empty_like(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: fill(_new_empty(self, *a, **k))


FIRST = []


def _trace_ops():
    import torch
    from speechdrivestemplates_amd import ops

    def tensors(x):
        if torch.is_tensor(x):
            return [x]
        if isinstance(x, (tuple, list)):
            return [t for y in x for t in tensors(y)]
        return []

    def bad(ts):
        out = []
        for i, t in enumerate(ts):
            if t.is_cuda and t.is_floating_point() and t.numel() and not torch.isfinite(t.float()).all().item():
                out.append((i, tuple(t.shape), str(t.dtype), int((~torch.isfinite(t.float())).sum().item())))
        return out

    def wrap(cls, which):
        orig = getattr(cls, which)

        def f(ctx, *a):
            out = orig(ctx, *a)
            if not FIRST:
                torch.cuda.synchronize()
                b = bad(tensors(out)) + ([("in",) + x for x in bad(tensors(a))] if which == "backward" else [])
                if b:
                    FIRST.append({"op": cls.__name__, "dir": which, "bad_outputs": b, "bad_inputs": bad(tensors(a)), "errs": repr(ops.streamk_error_codes())})
            return out

        setattr(cls, which, staticmethod(f))

    # the first encoder block's backward writes its weight gradient straight into the flat buffer (no output tensor): look at everything it touches
    l0_bwd = ops.L0BlockFn.backward

    def l0_backward(ctx, gz):
        out = l0_bwd(ctx, gz)
        if not any(f.get("op") == "L0BlockFn.inner" for f in FIRST):
            torch.cuda.synchronize()
            mel, w, mean, rstd, gamma, beta, mom = ctx.saved_tensors
            seen = {"gz": gz, "mel": mel, "w": w, "mean": mean, "rstd": rstd, "mom": mom, "gw": ops.grad_buffer(w)}
            b = {k: int((~torch.isfinite(v.float())).sum().item()) for k, v in seen.items() if v is not None}
            if any(b.values()):
                FIRST.append({"op": "L0BlockFn.inner", "nonfinite": b, "rstd_max": float(rstd.abs().max()), "mom_absmax": float(mom.abs().max())})
        return out

    ops.L0BlockFn.backward = staticmethod(l0_backward)
    from speechdrivestemplates_amd import dp
    launch0, wait0 = dp.GradReducer.launch, dp.GradReducer.wait

    def launch(self, opt, lo=0, hi=None):
        if self.active and (hi is None or hi > lo):
            torch.cuda.synchronize()
            seg = opt.flat_grad[lo:hi]
            nb = int((~torch.isfinite(seg)).sum().item())
            if nb:
                FIRST.append({"op": "GradReducer.launch:before", "lo": lo, "hi": hi, "nonfinite": nb})
            self._hunt = getattr(self, "_hunt", []) + [(opt, lo, hi)]
        return launch0(self, opt, lo, hi)

    def wait(self):
        wait0(self)
        torch.cuda.synchronize()
        for opt, lo, hi in getattr(self, "_hunt", []):
            nb = int((~torch.isfinite(opt.flat_grad[lo:hi])).sum().item())
            if nb:
                FIRST.append({"op": "GradReducer.wait:after", "lo": lo, "hi": hi, "nonfinite": nb})
        self._hunt = []

    dp.GradReducer.launch, dp.GradReducer.wait = launch, wait
    for name in dir(ops):
        cls = getattr(ops, name)
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
            wrap(cls, "forward")
            wrap(cls, "backward")


STASH = {}


def _stash_ops():
    """No host synchronisation, no timing change to speak of: keep references to what the first block's backward consumed and a device-side copy of
    the late gradient range as it was handed to the exchange; looked at only after a step came out non-finite."""
    import torch
    from speechdrivestemplates_amd import dp, ops
    l0_bwd = ops.L0BlockFn.backward
    take0 = ops._ARENA.take

    def l0_backward(ctx, gz):
        taken = []
        ops._ARENA.take = lambda n, dev: (taken.append(take0(n, dev)), taken[-1])[1]
        try:
            out = l0_bwd(ctx, gz)
        finally:
            ops._ARENA.take = take0
        mel, w, mean, rstd, gamma, beta, mom = ctx.saved_tensors
        STASH["l0"] = {"gz": gz, "mel": mel, "w": w, "mean": mean, "rstd": rstd, "mom": mom, "sums": taken[0] if taken else None,
                       "gw_after": ops.weight_storage(ops.grad_buffer(w)).detach().clone(), "groups": ctx.groups, "slope": ctx.slope}
        return out

    ops.L0BlockFn.backward = staticmethod(l0_backward)
    launch0 = dp.GradReducer.launch

    def launch(self, opt, lo=0, hi=None):
        if self.active and lo == 0 and hi is not None and hi <= 4096:
            STASH["pre_exchange"] = opt.flat_grad[lo:hi].detach().clone()
        return launch0(self, opt, lo, hi)

    dp.GradReducer.launch = launch


def _stash_report():
    import torch
    from speechdrivestemplates_amd import _lib, ops
    out = {}
    nf = lambda t: None if t is None else int((~torch.isfinite(t.double())).sum().item())  # noqa: E731
    if "pre_exchange" in STASH:
        out["pre_exchange_nonfinite"] = nf(STASH["pre_exchange"])
    d = STASH.get("l0")
    if d:
        out["l0_nonfinite"] = {k: nf(v) for k, v in d.items() if torch.is_tensor(v)}
        out["l0_absmax"] = {k: float(v.double().abs().max()) for k, v in d.items() if torch.is_tensor(v) and k in ("mom", "sums", "rstd", "mean", "gz")}
        # the same launch again, now alone on the GPU, into scratch accumulators
        lib = _lib.load()
        mel, w, gz = d["mel"], d["w"], d["gz"].contiguous()
        B, H, W = mel.shape
        sums = torch.zeros(11 * d["groups"] * 64, device=mel.device, dtype=torch.float64)
        gw = torch.zeros_like(ops.weight_storage(w))
        ops.check(lib.sdt_l0_block_bwd_t(gz.data_ptr(), ops._dt(gz), mel.data_ptr(), ops.weight_storage(w).data_ptr(), d["mean"].data_ptr(), d["rstd"].data_ptr(),
                                         None, None, d["mom"].data_ptr(), sums.data_ptr(), gw.data_ptr(), None, None, B, H, W, d["groups"], d["slope"], ops._stream()))
        torch.cuda.synchronize()
        out["l0_rerun_nonfinite"] = {"gw": nf(gw), "sums": nf(sums)}
        out["l0_rerun_vs_first"] = float((gw.reshape(-1) - d["gw_after"].reshape(-1)).abs().max()) if nf(d["gw_after"]) == 0 else "first was non-finite"
    return out


def _worker(rank, world, port, q, a):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if a.poison:
        _poison_allocations()
    from oracle import sdt_oracle as O
    from speechdrivestemplates_amd import dp, ops
    from speechdrivestemplates_amd.graph import GraphedStep
    from test_dp_gpu import _slice
    from test_model_gpu import _make_pipeline
    if a.trace:
        _trace_ops()
    else:
        _stash_ops()
    ops.set_storage(a.storage)
    if world > 1 and a.reserve >= 0:
        dp.RESERVED_SLOTS = a.reserve
        ops.SK_RESERVED_SLOTS_FWD = a.reserve
    pipe, _ = _make_pipeline(a.config, 16, 0.0)
    dev = torch.device("cuda", 0)
    gs = GraphedStep(pipe, warmup=1) if a.graph else None
    report = {"rank": rank, "events": []}
    names = {}
    for oname, opt in pipe.optimizers.items():
        off = 0
        spans = []
        for p in opt.params if hasattr(opt, "params") else []:
            spans.append((off, off + p.numel()))
            off += p.numel()
        names[oname] = spans
    for step in range(a.steps):
        full = O.make_batch(a.batch * world, 16, step=step, seed=1)
        batch = full if world == 1 else _slice(full, rank * a.batch, (rank + 1) * a.batch)
        batch = {k: (v.to(dev) if torch.is_tensor(v) and k != "num_frames" else v) for k, v in batch.items()}
        batch["speaker_stat"] = {k: v.to(dev) for k, v in batch["speaker_stat"].items()}
        if gs is not None:
            losses = gs.run(batch)
        else:
            losses, _ = pipe.forward_backward(batch)
            pipe.optimizer_updates(losses)
        torch.cuda.synchronize()
        ev = {"step": step, "errs": repr(ops.streamk_error_codes())}
        lbad = [k for k, v in losses.items() if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v).all().item()]
        if lbad:
            ev["losses"] = lbad
        for oname, opt in pipe.optimizers.items():
            for what in ("flat_grad", "flat_param"):
                t = getattr(opt, what)
                nb = int((~torch.isfinite(t)).sum().item())
                if nb:
                    idx = (~torch.isfinite(t)).nonzero().reshape(-1)
                    ev["%s.%s" % (oname, what)] = {"n": nb, "of": t.numel(), "first": int(idx[0]), "last": int(idx[-1])}
        if len(ev) > 2 or ev["errs"] != "{}":
            if not a.trace:
                try:
                    ev["stash"] = _stash_report()
                except Exception as e:  # noqa: BLE001
                    ev["stash"] = "report failed: %r" % (e,)
            report["events"].append(ev)
            break
    report["first_op"] = FIRST[:6]
    report["loss_hist_ok"] = True
    q.put(report)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--storage", default="bf16")
    ap.add_argument("--config", default="voice2pose_sdt_vae")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--reserve", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    import torch.multiprocessing as mp
    from test_dp_gloo import _collect, _free_port
    ctx = mp.get_context("spawn")
    nbad = 0
    for it in range(a.iters):
        t0 = time.time()
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, a.world, port, q, a)) for r in range(a.world)]
        for p in procs:
            p.start()
        try:
            res = sorted(_collect(procs, q, len(procs), 900), key=lambda r: r["rank"])
        except Exception as e:  # a worker died
            res = [{"rank": -1, "events": [{"died": repr(e)[:300]}]}]
        for p in procs:
            p.join(120)
        dirty = any(r["events"] or r.get("first_op") for r in res)
        nbad += bool(dirty)
        print("iter %d (%.1f s): %s" % (it, time.time() - t0, "NON-FINITE / ERROR: " + json.dumps(res) if dirty else "clean"), flush=True)
    print("hunt done: %d of %d iterations dirty (%s)" % (nbad, a.iters, vars(a)))


if __name__ == "__main__":
    main()
