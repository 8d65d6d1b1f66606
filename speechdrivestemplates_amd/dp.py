"""Data-parallel gradient exchange for the flat-buffer optimisers: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference wraps the model in DistributedDataParallel (core/pipelines/voice2pose.py:223): gradients are
summed over ranks in 25 MB buckets during backward and divided by world_size.  Here every optimiser group
already owns ONE contiguous gradient buffer (optim.FlatAdam), so the exchange is one summing all-reduce per
group -- 28.3 MB for the sdt_bp generator -- issued on a side stream as soon as that group's backward is
complete, and the 1/world_size is folded into the Adam kernel (``grad_scale``).  xGMI is point-to-point
(7 links/GPU): few large messages keep every link busy; there is nothing to gain from DDP's many small buckets.
"""
import os

import torch
import torch.distributed as dist


# Workgroup slots (of 512) that backward stream-K launches leave free while a gradient exchange can be in flight.  Measured on one GPU with an
# emulated collective (tools/debug/comm_emulation.py, profiles/r03_comm_emulation.txt: 32 workgroups holding their slots for 0.6 / 1.2 ms of each
# step): 7.67 -> 7.89 / 8.12 ms per step without a reserve, 7.8 -> 7.9 with 32 (64 buys nothing more); the reserve costs ~0.1 ms when nothing else
# runs, which is why it is not the single-GPU default.
RESERVED_SLOTS = 32


def reserved_slots():
    """The reserve a new GradReducer takes: RESERVED_SLOTS (32: what RCCL's all-reduce kernels were emulated with), or -- when the launcher caps or
    raises RCCL's channel count (NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS: one long-lived 256-thread workgroup per channel) -- that many slots,
    rounded up to the plan builder's multiple of 8 and kept within [8, 256]."""
    n = int(RESERVED_SLOTS)
    try:
        ch = max(int(os.environ.get("NCCL_MAX_NCHANNELS", 0)), int(os.environ.get("NCCL_MIN_NCHANNELS", 0)))
    except ValueError:
        ch = 0
    if ch > 0 and RESERVED_SLOTS == 32:  # (a caller that set RESERVED_SLOTS itself keeps its value)
        n = ch
    return max(8, min(256, (n + 7) // 8 * 8))


def other_gpu_processes():
    """PIDs (as the kernel driver numbers them) of OTHER processes that hold a compute context on a GPU of this node, from /sys/class/kfd/kfd/proc.
    The persistent kernels assume the workgroup slots they plan for are theirs: a second tenant (another rank on the same GPU, a profiler's agent
    process, somebody else's job) does not break results -- a starved launch sets its error word and the Trainer raises -- but it is worth a warning
    before the first step rather than an exception at the first log step.  Empty when the directory is not readable (containers often hide it)."""
    root = "/sys/class/kfd/kfd/proc"
    try:
        pids = [int(p) for p in os.listdir(root) if p.isdigit()]
    except OSError:
        return []
    me = os.getpid()
    return sorted(p for p in pids if p != me)
# Reducers of this process that currently hold the reserve: the knob is process-wide, so it is set when the FIRST active reducer
# appears and given back when the LAST one is closed -- a reducer dropped out of order (or collected late by the GC) must not switch the
# reserve off under one that is still exchanging gradients.
_RESERVE_HOLDERS = 0
_RESERVE_PREV = 0


def _acquire_reserve():
    global _RESERVE_HOLDERS, _RESERVE_PREV
    from . import ops
    if _RESERVE_HOLDERS == 0:
        _RESERVE_PREV = ops.SK_RESERVED_SLOTS
        ops.SK_RESERVED_SLOTS = reserved_slots()
    _RESERVE_HOLDERS += 1


def _release_reserve():
    global _RESERVE_HOLDERS
    from . import ops
    if _RESERVE_HOLDERS > 0:
        _RESERVE_HOLDERS -= 1
        if _RESERVE_HOLDERS == 0:
            ops.SK_RESERVED_SLOTS = _RESERVE_PREV


def active_reducers():
    """number of live GradReducers that overlap a gradient exchange with backward on this process's GPU"""
    return _RESERVE_HOLDERS


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_tensors(tensors, src=0):
    """Overwrite every tensor of ``tensors`` on every rank with rank ``src``'s contents.  Tensors of one dtype travel in
    one flattened message (DDP's constructor does the same with its coalesced ``_sync_params_and_buffers``)."""
    if world_size() == 1:
        return 0
    by_dtype = {}
    for t in tensors:
        if t is not None and t.numel():
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
    n = 0
    for (_dtype, _dev), group in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        dist.broadcast(flat, src)
        off = 0
        with torch.no_grad():
            for t in group:
                t.copy_(flat[off:off + t.numel()].view(t.shape))  # works for any stride pattern (weights live (Cout,*k,Cin))
                off += t.numel()
        n += flat.numel()
    return n


def replica_state(model, optimizers):
    """Everything that defines a replica: the optimisers' flat parameter / moment / step / lr buffers, every parameter that is
    not a view into one of them (the no-grad pose encoder, a frozen code table), every registered buffer (BatchNorm running
    statistics and counters, the mel window / filterbank, pose2pose's per-clip code buffers) and plain-tensor attributes such as
    external clip codes."""
    out, owned = [], set()
    for opt in optimizers:
        out += [opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.state_dev, opt.lr_dev]
        lo = opt.flat_param.data_ptr()
        owned.add((lo, lo + opt.flat_param.numel() * 4))
    for p in model.parameters():
        if not any(lo <= p.data_ptr() < hi for lo, hi in owned):
            out.append(p.data)
    out += list(model.buffers())
    ext = getattr(model, 'clips_code', None)
    if torch.is_tensor(ext) and not isinstance(ext, torch.nn.Parameter):
        # external clip codes (EXTERNAL_CODE configs, voice2pose.py:40-48) arrive as a host tensor and are moved to the GPU lazily
        # on first use; RCCL cannot broadcast a host tensor ("No backend type associated with device type cpu"), so they move now
        ref = out[0] if out else None
        if ref is not None and ext.device != ref.device:
            mover = getattr(model, '_code_table', None)
            ext = mover(ref.device) if mover is not None else ext.to(ref.device)
            model.clips_code = ext
        out.append(ext)
    return out


def sync_replicas(model, optimizers, src=0):
    """DistributedDataParallel's constructor semantics (reference core/pipelines/voice2pose.py:222-223, pose2pose.py:102):
    rank ``src``'s parameters and buffers -- and here also the Adam state, so a resumed run is consistent too -- replace every
    other rank's.  Without it ranks that were not seeded identically would apply the same averaged gradient to different
    weights for the whole run.  Returns the number of elements sent."""
    if world_size() == 1:
        return 0
    n = broadcast_tensors(replica_state(model, optimizers), src)
    for opt in optimizers:
        opt.mirrors.mark_dirty()  # weights were rewritten behind the mirrors' back
        opt._lr_host = float(opt.lr_dev.item())
        opt.param_groups[0]['lr'] = opt._lr_host
    return n


def sync_buffers(model, src=0):
    """DDP's ``broadcast_buffers=True`` re-sends rank 0's buffers at the start of EVERY forward.  This engine
    keeps buffers rank-local during training steps -- no collective on the step's critical path -- and calls this where the
    difference would otherwise be observable: before validation / test (eval-mode BatchNorm reads the running statistics) and
    before a checkpoint is written.  Rank 0 never receives in either scheme, so what rank 0 holds and saves is bit-identical
    to DDP's: its own batches' BatchNorm statistics and, for pose2pose, its own clips' codes (the reference's other ranks'
    writes to ``clip_code_mu/logvar`` are overwritten by the next broadcast, pose2pose.py:135-137)."""
    return broadcast_tensors(list(model.buffers()), src)


class GradReducer:
    def __init__(self, optimizers, overlap=True):
        self.optimizers = list(optimizers)
        self.ws = world_size()
        # SDT_DP_FORCE=1 exercises the whole exchange path on a single rank (a 1-rank all-reduce); used by the tests
        self.active = self.ws > 1 or (os.environ.get("SDT_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized())
        self._launched = {}  # id(opt) -> list of (lo, hi) ranges of flat_grad already in flight this step
        for opt in self.optimizers:
            opt.grad_scale = 1.0 / self.ws
        self.comm_stream = None
        if self.active and overlap and torch.cuda.is_available() and self.optimizers[0].flat_grad.is_cuda:
            self.comm_stream = torch.cuda.Stream()
        if self.active and torch.cuda.is_available() and self.optimizers[0].flat_grad.is_cuda:
            # the exchange's kernels (RCCL: a few dozen workgroups that live for a whole all-reduce) overlap the Conv2d backward, whose
            # persistent stream-K launches would otherwise occupy every workgroup slot of the GPU: leave the collective room (ops.SK_RESERVED_SLOTS)
            _acquire_reserve()
            self._holds_reserve = True
            self._startup_warnings()
        self._pending = []
        # graph.GraphedStep, "split" form (a backend whose collectives cannot be captured): while a step is being captured, all_reduce() hands the
        # exchange to this callback -- it closes the graph segment, records the exchange as an eager item and opens the next segment -- and the
        # bucket hooks of the backward pass stay quiet (one exchange per optimiser group, issued between the segments at replay)
        self.capture_cut = None
        self.launch_early = True  # False: the bucket hooks stay quiet, all_reduce() sends each group in one piece after backward (bench.py --dp-no-overlap)
        self._in_all_reduce = False
        # bench.py: pairs of events on the MAIN stream around all_reduce() -- the window is what the exchange costs the step
        # (launching the late buckets + waiting for the communication stream), i.e. the all-reduce time that backward did not hide
        self.exposed_events = None

    def _startup_warnings(self):
        """Things that cost a data-parallel run speed or robustness and are invisible otherwise: said once, up front."""
        import warnings
        from . import hw_queues
        val, late = hw_queues()
        if late:
            warnings.warn("GPU_MAX_HW_QUEUES=8 was set by `import speechdrivestemplates_amd` AFTER the HIP runtime had initialised (a torch.cuda call came "
                          "first): the runtime keeps its default of 4 hardware queues and the weight-gradient side stream will share a queue with the "
                          "main stream (~8 % of a data-parallel step).  Export GPU_MAX_HW_QUEUES=8 in the launcher or import this package first.",
                          RuntimeWarning, stacklevel=3)
        if self.ws > 1 and dist.get_backend() == "nccl" and not (os.environ.get("NCCL_MAX_NCHANNELS") or os.environ.get("NCCL_MIN_NCHANNELS")):
            warnings.warn("NCCL_MAX_NCHANNELS is not set: RCCL picks its own channel count (one long-lived workgroup per channel) while the backward "
                          "stream-K plans leave %d workgroup slots free for it; with more channels than that, persistent conv launches wait for slots a "
                          "collective holds.  Export NCCL_MAX_NCHANNELS=%d before init_process_group (bench.py does), or set dp.RESERVED_SLOTS to "
                          "RCCL's channel count." % (reserved_slots(), reserved_slots()), RuntimeWarning, stacklevel=3)
        others = other_gpu_processes()
        if others and self.ws == 1:  # (with world > 1 the other ranks of this node are expected to show up here, one per GPU)
            warnings.warn("%d other process(es) hold a compute context on this node's GPUs (kfd pids %s): the persistent conv kernels plan for a GPU of "
                          "their own apart from %d reserved workgroup slots; if one of them runs on THIS GPU expect starved launches (error words -> "
                          "RuntimeError at the next log step) -- raise dp.RESERVED_SLOTS (up to 256 = half of the GPU) for a shared device."
                          % (len(others), others[:8], reserved_slots()), RuntimeWarning, stacklevel=3)

    def close(self):
        """Give the process-wide plan knob back (a reducer that is torn down must not leave later single-GPU work planning with a reserve).
        Idempotent; counted per process (``_acquire_reserve``): the reserve goes away with the LAST active reducer, in whatever order they close.
        Pipelines call this explicitly (Trainer.close, a second setup_optimizer); ``__del__`` is only the safety net."""
        if getattr(self, "_holds_reserve", False):
            self._holds_reserve = False
            _release_reserve()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exposed_us(self):
        """mean main-stream time per all_reduce() call since ``exposed_events`` was set to a list (None: not recorded)"""
        if not self.exposed_events:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.exposed_events) * 1e3 / len(self.exposed_events)

    def launch(self, opt, lo=0, hi=None):
        """Start the summing all-reduce of ``opt.flat_grad[lo:hi]`` (call as soon as that range's backward kernels have
        been enqueued): it runs on the communication stream, behind everything queued on the main stream so far and
        concurrently with whatever backward work follows."""
        if not self.active:
            return
        from . import ops
        capturing = opt.flat_grad.is_cuda and torch.cuda.is_current_stream_capturing()
        if capturing and self.capture_cut is not None:
            return  # split-graph capture: the whole group goes out in all_reduce(), between the graph segments
        if not self.launch_early and not self._in_all_reduce:
            return
        hi = opt.flat_grad.numel() if hi is None else hi
        if hi <= lo:
            return
        self._launched.setdefault(id(opt), []).append((lo, hi))
        buf = opt.flat_grad[lo:hi]
        # side-stream weight-gradient kernels write into this buffer: the exchange waits for them -- on the communication
        # stream when there is one (the main stream keeps running backward), else by joining the main stream
        if self.comm_stream is not None:
            ops.flush_deferred_dw()
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            side = ops.side_stream_if_any()
            # inside a capture the side stream carries work only when ops.CAPTURE_SIDE_STREAMS forks the weight-gradient launches onto it
            # (then it is part of the capture and must be joined like in eager mode: ADVICE r5); otherwise it is idle and NOT capturing --
            # waiting on it would drag an uncaptured stream into the graph
            if side is not None and (not capturing or ops._side_ok()):
                self.comm_stream.wait_stream(side)
            with torch.cuda.stream(self.comm_stream):
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        else:
            ops.join_side_stream()
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        self._pending.append(work)

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def all_reduce(self, opts=None):
        """Exchange whatever part of each group's gradients has not been launched early, then wait for all of it."""
        on_gpu = self.optimizers[0].flat_grad.is_cuda
        capturing = on_gpu and torch.cuda.is_current_stream_capturing()
        if capturing and self.capture_cut is not None and self.active:
            group = list(opts) if opts is not None else None
            self.capture_cut(lambda: self.all_reduce(group))
            return
        timed = self.exposed_events is not None and self.active and on_gpu and not capturing
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        self._in_all_reduce = True
        try:
            for opt in (opts if opts is not None else self.optimizers):
                early = sorted(self._launched.pop(id(opt), []))
                pos, n = 0, opt.flat_grad.numel()
                for lo, hi in early + [(n, n)]:
                    if lo > pos:
                        self.launch(opt, pos, lo)
                    pos = max(pos, hi)
                self._launched.pop(id(opt), None)
        finally:
            self._in_all_reduce = False  # (a raising collective must not leave the bucket hooks of later steps in "late" mode)
        self.wait()
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.exposed_events.append((e0, e1))


def reduce_scalars(tensor_dict, dst=0, error_flag=None):
    """Trainer.reduce_tensor_dict (core/pipelines/trainer.py:323-327) with ONE packed collective instead of one blocking 4-byte collective per
    key; rank ``dst`` ends up with the mean over ranks.
    ``error_flag`` (a device scalar, ops.kernel_error_flag()): rides in the same message, which then is an all-reduce (same size, same cost), and
    the function returns the number of ranks that raised it -- on EVERY rank, so that all of them stop at the same step when one rank's
    persistent launch lost a partner (its NaN tile has reached every rank's gradients through the summing exchange).
    Without ``error_flag`` it returns the dict, as before."""
    ws = world_size()
    if ws == 1:
        return tensor_dict if error_flag is None else float(error_flag.item() != 0)
    keys = sorted(tensor_dict)
    vals = [tensor_dict[k].detach().double().reshape(()) for k in keys]
    if error_flag is not None:
        vals.append(error_flag.detach().double().reshape(()).to(vals[0].device) if vals else error_flag.detach().double().reshape(()))
    packed = torch.stack(vals)
    if error_flag is None:
        dist.reduce(packed, dst)
    else:
        dist.all_reduce(packed)
    if dist.get_rank() == dst:
        for i, k in enumerate(keys):
            tensor_dict[k] = (packed[i] / ws).to(tensor_dict[k].dtype)
    if error_flag is None:
        return tensor_dict
    return float(packed[-1].item())


def any_rank_flag(flag):
    """number of ranks on which the device scalar ``flag`` is non-zero, known on every rank (one 8-byte all-reduce)"""
    if world_size() == 1:
        return float(flag.item() != 0)
    t = flag.detach().double().reshape(1).ne(0).double()
    dist.all_reduce(t)
    return float(t.item())
