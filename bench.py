#!/usr/bin/env python
"""Benchmark of the SDT voice2pose training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (runs as typed: for N>1 without a launcher it starts its own N ranks,
                                                            one per GPU, as the reference does with mp.spawn, main.py:60-69;
                                                            under torch.distributed.run it takes the launcher's environment)

Workload (BASELINE.json configs[1]): voice2pose_sdt_bp, 32 clips per GPU (weak scaling), 64-frame / 121-kpt clips
(137 on-disk keypoints), fp32, synthetic seeded clips pre-staged in HBM.  One step = Voice2Pose.train_step without
logging: mel -> generator -> L1 + clip-code KL -> no-grad pose encoder x2 -> float64 metrics -> backward ->
gradient all-reduce (N>1) -> Adam on the clip-code table and the generator.  Prints ONE JSON line on rank 0.

Timing: W untimed warm-up steps, then exactly K steps between barrier + synchronize pairs; `value` = clips of all ranks / the
slowest rank's wall time (the driver's contract).  Every step is also bracketed by one HIP event on the main stream, so the
line carries the MEDIAN step time (SURVEY.md 8d's definition) and the mean over the steps that carried no per-kernel events.

`roofline`        : the dominant kernel (the split-fp32 implicit-GEMM conv; with --no-f32-split the fp32-MFMA one): algorithmic FLOPs
                    per launch / mean launch duration from HIP events on the launching stream over sampled steps of the timed
                    region, against ITS matrix peak -- six bf16 MFMA products per fp32 product: the dense bf16 peak / 6 = 416.7
                    TFLOP/s of algorithmic fp32 FLOPs (fp32-MFMA kernels: 157.3).  `traffic` is NOT measured in this process (PMC counters need rocprofv3):
                    the line cites the committed measurement and says which source revision it was taken at, or null when the
                    kernel source has changed since.
`roofline_conv1d` : the Conv1d stacks (U-Net + decoder forward/backward, the two pose-encoder passes, their weight
                    gradients): algorithmic bytes and FLOPs of the stage / the stage's event-timed windows,
                    against 8 TB/s and 157.3 TFLOP/s -- the north-star's ">= 40 % HBM roofline on the Conv1d stacks" tracker.
`cpu_baseline`    : the CPU oracle (PyTorch-CPU restatement of the reference step, oracle/) timed on this box's host cores for
                    a few steps of the same workload (rank 0, N=1 only).
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime initialises (speechdrivestemplates_amd/__init__.py says why)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
# the split-fp32 conv kernel (csrc/convbf.hip, ET = float): every fp32 product = six bf16 MFMA products, so its matrix roofline in ALGORITHMIC
# (fp32) FLOP/s is a sixth of the dense bf16 peak
SPLIT_F32_PEAK_TFLOPS = BF16_MATRIX_PEAK_TFLOPS / 6.0


def matrix_peak_tflops(kernel_name):
    return SPLIT_F32_PEAK_TFLOPS if kernel_name.startswith(("convbf2_kernel<float", "convx3_dw_kernel")) else FP32_MATRIX_PEAK_TFLOPS
HBM_PEAK_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
N_CLIPS = 4096
# Algorithmic work of the Conv1d stacks per 32-clip step (weights counted once per step): generator U-Net + decoder
# forward+backward 153.3 MB / 3 x 0.243 GFLOP per clip; two no-grad pose-encoder passes 40.6 MB / 2 x 0.0806 GFLOP per clip
CONV1D_MB_PER_32 = {"generator_fwd_bwd": 153.3, "pose_encoder_x2": 40.6}
CONV1D_GFLOP_PER_CLIP = {"generator_fwd_bwd": 3 * 0.243, "pose_encoder_x2": 2 * 0.0806}
# timed steps that carry HIP events around every conv launch, half of them "alone", half "as run" (see below).  A sampled
# step costs ~1.6 ms extra (the events serialise neighbouring launches, and an "alone" step gives up the side stream), so they
# are kept to ~1 in 15
EVENT_STEPS_MAX = int(os.environ.get("SDT_EVENT_STEPS", "6"))


def event_steps(steps):
    """sampled steps of the timed region: below 16 steps one "alone" + one "stages only"; from 16 steps on THREE "alone" steps + one
    "stages only" (one sampled step of 20 was a thin basis for roofline.achieved) -- the first with events around every conv
    launch, the other two around the dominant kernel's launches only (a fully instrumented step costs ~0.7 ms; three of them were 2 % of the
    driver's 20-step run); a rotation alone / stages / as-run from 40 steps on, two rotations from 60 on"""
    if steps < 2:
        return steps  # a single timed step still carries the roofline sample
    if steps < 16:
        return 2
    return 4 if steps < 40 else min(EVENT_STEPS_MAX, (steps // 30) * 3 if steps >= 60 else 3)


def stage_batches(n_batches, B, rank, dev):
    """Seeded synthetic batches with the DataLoader-collated layout, resident on the device."""
    import numpy as np
    out = []
    for i in range(n_batches):
        rng = np.random.Generator(np.random.PCG64([1, rank, i]))
        idx = (np.arange(B) + (rank * n_batches + i) * B) % N_CLIPS
        out.append({
            "audio": torch.from_numpy((0.1 * rng.standard_normal((B, 68266))).astype(np.float32)).to(dev),
            "poses": torch.from_numpy(rng.standard_normal((B, 64, 2, 121)).astype(np.float32)).to(dev),
            "clip_index": torch.from_numpy(idx.astype(np.int64)).to(dev),
            "num_frames": torch.full((B,), 64, dtype=torch.int64),  # host tensor: only [0] is read
            "speaker": ["synthetic"] * B,
            "speaker_stat": {"scale_factor": torch.from_numpy(rng.uniform(0.8, 1.3, (B,))).to(dev),
                             "mean": torch.from_numpy(rng.standard_normal((B, 242)) * 20.0).to(dev),
                             "std": torch.from_numpy(rng.uniform(2.0, 30.0, (B, 242))).to(dev)},
        })
    return out


def cpu_baseline(B, steps=3):
    """The CPU restatement of the same train step (oracle/, test infrastructure) on the host cores."""
    from oracle import sdt_oracle as O
    cores = os.cpu_count() or 1
    phys = max(1, cores // 2)  # SMT siblings do not help MKL-DNN convolutions
    cfg = O.cfg_named("voice2pose_sdt_bp")
    eng = O.OracleVoice2Pose(cfg, O.make_voice2pose_state(cfg, N_CLIPS, seed=0, code_std=0.5))
    batches = [O.make_batch(B, N_CLIPS, step=i, seed=1) for i in range(2)]
    # the CPU path does not scale to every core of a 2-socket host: sweep a few thread counts (1 timed step each)
    # and report the best one, then time `steps` steps there -- the baseline gets its most favourable setting
    trial = {}
    for nt in sorted({min(phys, n) for n in (16, 32, 64, phys)}):
        torch.set_num_threads(nt)
        eng.train_step(batches[0])  # warm-up (thread pool, primitive caches)
        t0 = time.perf_counter()
        eng.train_step(batches[1])
        trial[nt] = time.perf_counter() - t0
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(batches[i % 2])
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "clips/s", "cores": best, "kind": "port",
            "sample": "%d steps of %d clips, PyTorch-CPU fp32 oracle (oracle/sdt_oracle.py), best of thread sweep %s -> %d threads, %.2f s/step"
                      % (steps, B, {k: round(v, 2) for k, v in trial.items()}, best, dt / steps),
            "host_cores": phys, "cpu": _cpu_model(), "torch": torch.__version__}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_digest():
    """sha256 over the kernel sources: what a committed PMC measurement has to match to be cited."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "speechdrivestemplates_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def cited_traffic(kernel_name):
    """HBM-side traffic cannot be read from inside the process (PMC counters need rocprofv3, separate passes for FETCH_SIZE and
    WRITE_SIZE).  Cite the newest committed measurement of this command (tools/hbm_traffic.py) -- but only when it was taken on
    the kernel sources that are running now; otherwise report null and say why."""
    prof = os.path.join(REPO, "profiles")
    cands = sorted((f for f in os.listdir(prof) if f.endswith("_hbm_traffic_bench.json")), reverse=True) if os.path.isdir(prof) else []
    info = {"traffic": None, "traffic_measured_in_run": False}
    for f in cands:
        tj = json.load(open(os.path.join(prof, f)))
        if tj.get("dominant_kernel", "") != kernel_name:
            continue
        info["traffic_source"] = "profiles/" + f
        info["traffic_source_kernel_digest"] = tj.get("kernel_source_digest")
        info["traffic_source_git_head"] = tj.get("git_head")
        if tj.get("kernel_source_digest") == kernel_source_digest():
            info["traffic"] = tj["traffic_mb_per_launch"] * 1e6
            info["traffic_unit"] = "bytes/launch (2*FETCH_SIZE+WRITE_SIZE, fabric side, upper bound on HBM)"
        else:
            info["traffic_note"] = "kernel sources changed since that measurement (digest now %s): not cited" % kernel_source_digest()
        break
    return info


def cited_sustained_mhz():
    """Shader clock during the dominant kernel's launches (GRBM_GUI_ACTIVE / duration) from the newest committed PMC pass (tools/pmc_summary.py
    prints 'clock N GHz' per kernel); None when no summary carries it."""
    import re
    prof = os.path.join(REPO, "profiles")
    for f in sorted((f for f in os.listdir(prof) if "_pmc_" in f and f.endswith(".txt")), reverse=True) if os.path.isdir(prof) else []:
        m = re.search(r"clock ([0-9.]+) GHz", open(os.path.join(prof, f)).read())  # first line = the dominant kernel's launches
        if m:
            return {"sustained_mhz": float(m.group(1)) * 1e3, "sustained_mhz_source": "profiles/" + f}
    return {"sustained_mhz": None}


def cited_trace_fraction(kernel_name):
    """The rocprofv3 kernel-trace average of the dominant kernel (tools/collect_profiles.sh writes profiles/<round>_trace_fraction.json next
    to the trace summaries): cited beside the event-based figure while the kernel sources still hash to the digest recorded there."""
    prof = os.path.join(REPO, "profiles")
    cands = sorted((f for f in os.listdir(prof) if f.endswith("_trace_fraction.json")), reverse=True) if os.path.isdir(prof) else []
    for f in cands:
        tj = json.load(open(os.path.join(prof, f)))
        if tj.get("kernel") != kernel_name:
            continue
        ok = tj.get("kernel_source_digest") == kernel_source_digest()
        return {"source": "profiles/" + f, "avg_launch_us": tj.get("avg_launch_us") if ok else None, "frac": tj.get("frac") if ok else None,
                "current": ok, "note": None if ok else "kernel sources changed since that trace (digest now %s): not cited" % kernel_source_digest()}
    return None


class _StubPipeline:
    """SDT_BENCH_STUB=1: the control flow of this script (process group, barriers, timed loop, max over ranks, JSON) with the
    train step replaced by a tiny CPU all-reduce -- tests/test_bench_flow.py runs it under gloo with two ranks so that the
    driver's multi-GPU launch cannot die on plumbing that a 1-GPU box never executes."""

    def __init__(self, world):
        self.world = world
        self.buf = torch.ones(1024)

    def forward_backward(self, batch):
        time.sleep(0.002)
        return {"G_loss": torch.tensor(0.5)}, {}

    def optimizer_updates(self, losses):
        if self.world > 1:
            dist.all_reduce(self.buf)
            self.buf.div_(self.world)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--config", default="voice2pose_sdt_bp")
    ap.add_argument("--conv-math", default="f32", choices=["f32", "bf16x6", "bf16x3", "bf16"],
                    help="product arithmetic of the forward / input-gradient conv kernels (experiments; the metric is quoted on f32)")
    ap.add_argument("--storage", default="f32", choices=["f32", "bf16"],
                    help="element type of the Conv2d chain's activations and conv operands in HBM: bf16 = BASELINE config 4's arithmetic "
                         "(bf16 tensors and MFMA products, fp32 accumulation / statistics / master weights); the metric is quoted on f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the informational bf16-storage leg (BASELINE config 4's arithmetic) after the timed region")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-launch HIP events (roofline leg)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (N=1)")
    ap.add_argument("--graph-streams", action="store_true", help="with --graph: capture the side streams too (experiment)")
    ap.add_argument("--no-overlap-aux", action="store_true", help="keep the no-grad pose-encoder passes on the main stream")
    ap.add_argument("--no-stats-fusion", action="store_true", help="separate statistics pass for the 2-D norms (A/B of the fused conv epilogue)")
    ap.add_argument("--no-pair-aux", action="store_true", help="the two no-grad pose-encoder passes as two passes (A/B of the batched pass)")
    ap.add_argument("--no-group-dw", action="store_true", help="the deferred small weight gradients one launch + reduce per layer instead of one grouped launch (A/B)")
    ap.add_argument("--no-chain1d", action="store_true", help="the generator's Conv1d stage block by block (two to three launches each) instead of one persistent launch per direction (A/B)")
    ap.add_argument("--no-overlap-dw", action="store_true", help="keep the weight-gradient kernels on the main stream (default: side stream, +4 %)")
    ap.add_argument("--atomic-dw", action="store_true",
                    help="weight gradients with fp32 atomics (the round-2 default) instead of the ordered, bit-reproducible reductions")
    ap.add_argument("--bwd-wpc", default=None, help="experiment: persistent workgroups per CU of the backward stream-K plans, 'DX,DW' (e.g. 1,1 or 2,1)")
    ap.add_argument("--dp-legs", action="store_true", help="N > 1: also run the informational legs behind the timed region (graph replay of the data-parallel step, bf16 storage) -- "
                                                           "they issue collectives on every rank; off by default between real ranks so that nothing behind the timed region can take the scaling number with it")
    ap.add_argument("--dp-graph-full", action="store_true", help="N > 1: the informational graph-replay legs capture the RCCL all-reduces WITH the step (graph.GraphedStep "
                    "mode 'full', the product default under SYS.HIP_GRAPH; exercised here on a 1-rank group only) instead of graph segments around an eager exchange ('split')")
    ap.add_argument("--no-f32-split", action="store_true", help="Conv2d forward / input gradient on the fp32-MFMA kernels of rounds 3-4 (A/B of the split-fp32 kernel)")
    ap.add_argument("--no-streamk", action="store_true", help="all Conv2d launches on the 64x64 kernel of conv.hip (A/B of the persistent stream-K kernel)")
    ap.add_argument("--no-streamk-dw", action="store_true", help="weight gradients on the atomics kernel of conv.hip (A/B of the deterministic stream-K weight gradient)")
    ap.add_argument("--streamk-min-steps", type=int, default=None, help="experiment: K steps per tile from which a launch takes the stream-K kernel")
    ap.add_argument("--streamk-min-cout", type=int, default=None, help="experiment: GEMM width from which an input-gradient launch takes the stream-K kernel")
    ap.add_argument("--no-small1d", action="store_true", help="tuning library only (SDT_HIP_LIB): 1-D launches on conv_taps_kernel instead of conv1d_small_kernel (A/B)")
    ap.add_argument("--dp-reserve", type=int, default=None, help="experiment (N > 1 / SDT_DP_FORCE): workgroup slots the backward stream-K plans leave free for the collective (default dp.RESERVED_SLOTS)")
    ap.add_argument("--dp-no-overlap", action="store_true", help="experiment: no early bucket launches from the backward hooks -- one exchange per optimiser group after backward")
    ap.add_argument("--no-defer-dw", action="store_true", help="launch the 1-D stage's weight gradients inline (default: one batch on the side stream under the Conv2d backward)")
    args = ap.parse_args(argv)

    stub = os.environ.get("SDT_BENCH_STUB") == "1"
    backend = os.environ.get("SDT_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for the CPU control-flow test
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as a plain command: become the launcher (the reference spawns its own ranks too, main.py:60-69) -- re-exec under
        # torch.distributed.run with one rank per GPU on this node; rank 0 of the children prints the JSON line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = not stub
    # SDT_DP_FORCE=1 under a launcher (`torch.distributed.run --nproc-per-node 1`): a 1-rank RCCL group, so that the whole data-parallel step --
    # reducer, bucket hooks, reserve, graph capture of the all-reduces -- is what gets timed on a single-GPU box
    forced_dp = world == 1 and os.environ.get("SDT_DP_FORCE") == "1" and "WORLD_SIZE" in os.environ and not stub
    if world > 1 or forced_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            # RCCL runs one long-lived workgroup per channel; the backward stream-K plans leave dp.reserved_slots() workgroup slots free for them
            # (32, or this variable when the launcher sets it).  Unset, RCCL picks its own channel count for the topology -- possibly more than the
            # reserve, and the persistent kernels would then wait for slots held by a collective that waits for a slower rank.  Cap it to the reserve.
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
        if on_gpu:
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            try:  # device_id binds the communicator to this rank's GPU up front
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            except TypeError:
                dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, "--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)

    B = args.batch
    if stub:
        ops = None
        pipe = _StubPipeline(world)
        batches = [None]
    else:
        from __graft_entry__ import make_pipeline
        from speechdrivestemplates_amd import ops
        ops.OVERLAP_DW = not args.no_overlap_dw
        ops.USE_STREAMK = not args.no_streamk
        ops.F32_SPLIT = not args.no_f32_split
        ops.USE_STREAMK_DW = not args.no_streamk and not args.no_streamk_dw
        if args.streamk_min_steps is not None:
            ops.STREAMK_MIN_STEPS = args.streamk_min_steps
        if args.streamk_min_cout is not None:
            ops.STREAMK_MIN_COUT = args.streamk_min_cout
        if args.no_small1d:
            import ctypes
            from speechdrivestemplates_amd import _lib
            _lib.check(_lib.load().sdt_debug_set_small1d(ctypes.c_int(0)))
            ops.STREAMK_MIN_COUT = 64
        if args.bwd_wpc:
            ops.SK_WPC_DX, ops.SK_WPC_DW = (int(v) for v in args.bwd_wpc.split(","))
        ops.DEFER_SMALL_DW = not args.no_defer_dw
        ops.DETERMINISTIC_DW = not args.atomic_dw
        if args.atomic_dw:
            ops.USE_STREAMK_DW = False
        ops.PROFILER_NO_FUSION = bool(args.no_stats_fusion)
        ops.CHAIN1D = not args.no_chain1d
        ops.GROUP_DW = not args.no_group_dw
        ops.PAIR_AUX = not args.no_pair_aux
        ops.CAPTURE_SIDE_STREAMS = bool(args.graph_streams)
        ops.OVERLAP_AUX = not args.no_overlap_aux
        ops.set_conv_math(args.conv_math)
        ops.set_storage(args.storage)
        if args.dp_reserve is not None:
            from speechdrivestemplates_amd import dp as _dp
            _dp.RESERVED_SLOTS = args.dp_reserve
        # every rank draws its own initial weights (nothing here seeds torch): setup_optimizer's dp.sync_replicas makes the
        # replicas identical, as DDP's constructor does in the reference
        pipe, cfg = make_pipeline(args.config, N_CLIPS, batch_global=B * world,
                                  sys_opts={"STORAGE": args.storage, "CHAIN1D": not args.no_chain1d, "CONV_F32_SPLIT": not args.no_f32_split,
                                            "DISTRIBUTED": world > 1 or forced_dp})
        batches = stage_batches(4, B, rank, dev)
        if args.dp_no_overlap and getattr(pipe, "reducer", None) is not None:
            pipe.reducer.launch_early = False

    def step(i):
        losses, _ = pipe.forward_backward(batches[i % len(batches)])
        pipe.optimizer_updates(losses)
        return losses

    runner = step
    if args.graph and not stub:  # N > 1: the gradient exchange is part of the replayed step (graph.GraphedStep, "full" over RCCL)
        from speechdrivestemplates_amd.graph import GraphedStep
        gs = GraphedStep(pipe, warmup=min(3, max(1, args.warmup - 1)))
        runner = lambda i: gs.run(batches[i % len(batches)])  # noqa: E731

    # (the event pools are created BEFORE the warm-up: creating and recording ~5000 events takes the host ~50 ms, the GPU idles and drops
    # its clocks meanwhile, and the first six timed steps then ran 0.2-2 ms slow -- 2.5 % of the driver's 20-step run)
    prof = prof_ovl = prof_dom = stages = None
    if not stub and not args.no_kernel_events and not args.graph:
        prof = ops.ConvProfiler(pool=2 * 200 * EVENT_STEPS_MAX)
        prof_dom = ops.ConvProfiler(pool=2 * 40 * EVENT_STEPS_MAX)  # short runs: "alone" steps that time the dominant kernel's launches only
        prof_ovl = ops.ConvProfiler(pool=2 * 200 * EVENT_STEPS_MAX) if ops.OVERLAP_DW else None
        stages = ops.StageTimer()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if on_gpu else None
    if marks is not None:
        for e in marks:  # hipEventCreate happens at the first record()
            e.record()
    for i in range(args.warmup):
        runner(i)
    reducer = getattr(pipe, "reducer", None)
    if world > 1 and reducer is not None and on_gpu:
        reducer.exposed_events = []  # main-stream event windows around the gradient exchange of every timed step
    # HIP events around every conv launch on a few timed steps, recorded on the stream each kernel is launched on.  By default the
    # weight-gradient kernels run on a second stream, concurrently with the input-gradient chain, so a launch's duration
    # then includes the time it shared the GPU: sampled steps therefore ALTERNATE between "alone" (side stream off for
    # that step: the kernel-quality figure reported as roofline.achieved) and "as run" (roofline.overlapped).
    n_ev = event_steps(args.steps) if prof is not None else 0
    sampled = sorted({(j + 1) * args.steps // (n_ev + 1) for j in range(n_ev)}) if n_ev else []  # spread over the timed region
    # one event per step boundary on the main stream: per-step GPU time without a host synchronisation
    host_marks = []
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    overlap_dw = ops.OVERLAP_DW if ops is not None else False
    n_alone = n_ovl = n_stage = n_dom = 0
    for i in range(args.steps):
        if marks is not None:
            marks[i].record()
        else:
            host_marks.append(time.perf_counter())
        if prof is not None:  # sampled: the events serialise the host a little and cost a few % on the steps they cover
            ops.PROFILER, ops.STAGES, ops.OVERLAP_DW = None, None, overlap_dw
            if i in sampled:
                # three kinds of sampled step in rotation: per-launch events with the side stream off ("alone"), per-launch events
                # as run, and stage windows only (a handful of events per step: per-launch events would inflate the windows)
                short_run = 16 <= args.steps < 40  # three "alone" steps, then one "stages only"
                if short_run and 0 < n_alone and n_alone + n_dom < 3:
                    # the second and third "alone" step of a short run carry events around the dominant kernel's launches only: an event
                    # pair costs a launch its overlap with its neighbours (~0.5 ms per fully instrumented step, 2 % of a 20-step run)
                    if prof_dom.only is None:  # the kernel with the most algorithmic FLOPs on the first sampled step (no synchronisation here)
                        fl = {}
                        for rec in prof.records:
                            fl[rec[0]] = fl.get(rec[0], 0.0) + rec[3]
                        prof_dom.only = {max(fl, key=fl.get)} if fl else set()
                    ops.PROFILER, ops.OVERLAP_DW, n_dom = prof_dom, False, n_dom + 1
                elif (short_run and n_alone < 1) or (not short_run and n_alone <= n_ovl):
                    ops.PROFILER, ops.OVERLAP_DW, n_alone = prof, False, n_alone + 1
                elif n_stage < n_alone:
                    ops.STAGES, n_stage = stages, n_stage + 1
                elif prof_ovl is not None:
                    ops.PROFILER, n_ovl = prof_ovl, n_ovl + 1
                else:
                    ops.PROFILER, ops.OVERLAP_DW, n_alone = prof, False, n_alone + 1
        losses = runner(args.warmup + i)
    if marks is not None:
        marks[args.steps].record()
    else:
        host_marks.append(time.perf_counter())
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if ops is not None:
        ops.PROFILER, ops.STAGES, ops.OVERLAP_DW = None, None, overlap_dw
        if on_gpu and not stub:  # a stream-K launch that gave up waiting for a partial tile would have left a code (and wrong results)
            codes = ops.streamk_error_codes()
            assert not codes, "stream-K error words: %r" % codes
    prof_steps = n_alone
    if marks is not None:
        step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    else:
        step_ms = [1e3 * (host_marks[i + 1] - host_marks[i]) for i in range(args.steps)]
    clean_ms = [t for i, t in enumerate(step_ms) if i not in sampled] or step_ms
    per_rank = None
    if world > 1:
        exposed = reducer.exposed_us() if (reducer is not None and on_gpu) else None
        mine = torch.tensor([elapsed, statistics.median(step_ms), sum(clean_ms) / len(clean_ms), -1.0 if exposed is None else exposed],
                            device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rows = [[float(x) for x in g.tolist()] for g in gathered]
        per_rank = {"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                    "elapsed_s": [r[0] for r in rows], "median_ms_per_step": [r[1] for r in rows],
                    "exposed_allreduce_us_per_step": None if rows[0][3] < 0 else [r[3] for r in rows],
                    "note": "exposed_allreduce_us: main-stream HIP-event window around GradReducer.all_reduce (late buckets + wait for "
                            "the communication stream) -- the part of the gradient exchange that backward did not hide"}
        elapsed, median_ms, clean_mean_ms = (max(r[k] for r in rows) for k in range(3))  # the slowest rank defines the job
    else:
        median_ms, clean_mean_ms = statistics.median(step_ms), sum(clean_ms) / len(clean_ms)
    final_loss = float(losses["G_loss" if "G_loss" in losses else "loss"].detach())
    assert final_loss == final_loss and final_loss < 10.0, "training diverged: G_loss=%r" % final_loss

    # The informational legs behind the timed region must never take the bench line with them.  Between REAL ranks (never available to the build
    # container) they replay graph SEGMENTS around an eagerly issued exchange: the sequence of collectives is then the eager step's whatever happens
    # to a capture; capturing the all-reduces with the step ('full') is opt-in there (--dp-graph-full) and the default on one rank, where it is tested.
    leg_graph_mode = "split" if (world > 1 and not args.dp_graph_full) else None

    def graph_leg_f32():
        """N > 1 (or a forced 1-rank group), informational: the SAME fp32 step replayed from a hipGraph with its RCCL all-reduces captured
        (graph.GraphedStep): enqueued launch by launch the data-parallel step pays ~0.4-0.6 ms of host work for the bucket launches on top of the
        reserve (profiles/r05_dp_one_gpu_ab.txt: 7.80 vs 7.15 ms on one GPU; replayed 7.34).  `value` stays the eager step."""
        from speechdrivestemplates_amd.graph import GraphedStep
        try:
            gs = GraphedStep(pipe, warmup=1, mode=leg_graph_mode)
            base = args.warmup + args.steps
            for i in range(4):
                gs.run(batches[(base + i) % len(batches)])
            sync()
            if world > 1:
                dist.barrier()
            t0g = time.perf_counter()
            n = 20
            for i in range(n):
                lg = gs.run(batches[(base + 4 + i) % len(batches)])
            sync()
            ms = (time.perf_counter() - t0g) * 1e3 / n
            if world > 1 or forced_dp:
                t = torch.tensor([ms], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t.item())
            return {"value": world * B / (ms * 1e-3), "unit": "clips/s", "ms_per_step": ms, "steps": n, "graph_mode": gs.mode, "G_loss": float(lg["G_loss"].detach()),
                    "note": "the timed fp32 step replayed from a hipGraph (SYS.HIP_GRAPH) with the gradient exchange captured; not `value`"}
        except Exception as e:  # a capture that fails must not take the bench line with it
            return {"error": repr(e)[:300]}

    def bf16_leg():
        """informational, OUTSIDE the timed region and not part of `value`: BASELINE config 4's arithmetic on this GPU (these GPUs: every rank runs
        it, the exchange included; the slowest rank's time counts) -- the same train step with the Conv2d chain's tensors stored as bf16 and its
        products on the bf16 MFMA (fp32 accumulation / statistics / master weights; tests/test_bf16_gpu.py states and checks its tolerances).
        Replayed from a hipGraph (N > 1: with its RCCL all-reduces, graph.GraphedStep): the step needs ~2-3 ms of GPU time, the host ~3-5 ms to
        enqueue its launches one by one (tools/host_time.py)."""
        out = {}

        def wall(t_local):  # the slowest rank defines the job
            if world == 1 and not forced_dp:
                return t_local
            t = torch.tensor([t_local], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        from speechdrivestemplates_amd.graph import GraphedStep
        ops.set_storage("bf16")
        pipe.knobs["storage"] = "bf16"  # the SAME pipeline continues in bf16 storage (its bf16 weight copies are allocated on first use)
        try:
            base = args.warmup + args.steps
            for i in range(3):  # eager: allocates the bf16 weight copies and builds the bf16 plans outside any capture
                step(base + i)
            sync()
            aprof = None
            if prof is not None:  # its own roofline: three eager steps with per-launch events, weight gradients on the main stream
                aprof = ops.ConvProfiler(pool=2 * 200 * 3)
                try:
                    ops.PROFILER, ops.OVERLAP_DW = aprof, False
                    for i in range(3):
                        step(base + 3 + i)
                    sync()
                finally:
                    ops.PROFILER, ops.OVERLAP_DW = None, overlap_dw
            gs = GraphedStep(pipe, warmup=1, mode=leg_graph_mode)
            for i in range(4):
                gs.run(batches[(base + 6 + i) % len(batches)])
            sync()
            if world > 1:
                dist.barrier()
            ta = time.perf_counter()
            n_alt = 30
            for i in range(n_alt):
                losses_alt = gs.run(batches[(base + 10 + i) % len(batches)])
            sync()
            alt_ms = wall((time.perf_counter() - ta) * 1e3 / n_alt)
            alt_loss = float(losses_alt["G_loss"].detach())
            te = time.perf_counter()
            for i in range(10):
                step(base + 40 + i)
            sync()
            eager_ms = wall((time.perf_counter() - te) * 1e3 / 10)
            codes = ops.streamk_error_codes()
            assert not codes, "stream-K error words (bf16 leg): %r" % codes
        finally:
            ops.set_storage("f32")
            pipe.knobs["storage"] = "f32"
        assert alt_loss == alt_loss and alt_loss < 10.0, "bf16 leg diverged: G_loss=%r" % alt_loss
        out["alt_conv_math"] = {"mode": "bf16", "dtype": "bf16 (Conv2d chain: bf16 tensors in HBM + bf16 MFMA products; fp32 accumulation, statistics, "
                                                         "master weights, gradients; generator Conv1d chain: fp32 tensors, bf16 MFMA products; 1-D weight gradients, head, losses fp32)",
                                "value": world * B / (alt_ms * 1e-3), "unit": "clips/s", "n_gpus": world, "ms_per_step": alt_ms, "steps": n_alt, "graph": True,
                                "eager_ms_per_step": eager_ms, "G_loss": alt_loss, "graph_mode": gs.mode,
                                "note": "not the headline (the metric is quoted on the reference's fp32 arithmetic): %d further steps of the same run in "
                                        "bf16 storage (BASELINE config 4 per GPU), replayed from a hipGraph; eager_ms_per_step = the same steps "
                                        "enqueued launch by launch (host-bound)" % n_alt}
        if aprof is not None:
            asum = aprof.summary()
            two_d = {k: v for k, v in asum.items() if "bf16" in k or k.startswith("convbf")}
            aname, ad = max(two_d.items(), key=lambda kv: kv[1]["us"])
            a_us = ad["us"] / ad["launches"]
            a_bytes, a_flops = ad["bytes"] / ad["launches"], ad["flops"] / ad["launches"]
            tot_b = sum(v["bytes"] for v in two_d.values())
            tot_f = sum(v["flops"] for v in two_d.values())
            tot_us = sum(v["us"] for v in two_d.values())
            out["alt_conv_math"]["roofline"] = {
                "bound": "hbm", "kernel": aname, "achieved": a_bytes / (a_us * 1e-6) / 1e9, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                "frac": a_bytes / (a_us * 1e-6) / 1e12 / HBM_PEAK_TBS, "avg_launch_us": a_us, "launches_per_step": ad["launches"] / 3.0,
                "algorithmic_mb_per_launch": a_bytes / 1e6, "algorithmic_gflop_per_launch": a_flops / 1e9,
                "mfma_achieved_tflops": a_flops / (a_us * 1e-6) / 1e12, "mfma_frac": a_flops / (a_us * 1e-6) / 1e12 / BF16_MATRIX_PEAK_TFLOPS,
                "all_conv2d_launches": {"ms_per_step": tot_us / 3.0 / 1e3, "gb_s": tot_b / (tot_us * 1e-6) / 1e9,
                                        "tflops": tot_f / (tot_us * 1e-6) / 1e12, "launches_per_step": sum(v["launches"] for v in two_d.values()) / 3.0},
                "event_sampled_steps": 3,
                "note": "bf16 tensors: algorithmic bytes = 2 x (|X| + |Y| + |W|) per launch (SURVEY.md 8d's per-layer bytes halved) / HIP-event window, "
                        "against 8 TB/s; mfma_* = the same launches against the dense bf16 MFMA peak (%.0f TFLOP/s): the kernels are bound by "
                        "neither -- LDS feeding and the per-tile epilogue of a persistent fp32-shaped tile (DESIGN.md section 2)" % BF16_MATRIX_PEAK_TFLOPS}
        return out.get("alt_conv_math")

    def pcie_leg():
        """informational, OUTSIDE the timed region and never `value`: the reference's boundary hands over HOST tensors (voice2pose.py:86-90 moves
        the DataLoader's collated batch with .cuda()).  The same fp32 step fed from pinned host memory: (a) `prefetched` -- the next batch's copies
        are issued on a copy stream while the current step runs (DataLoader(pin_memory=True) + non_blocking copies); (b) `serial` -- the copies sit
        on the step's own stream in front of it.  One rank only: a measurement of the host link of this box, not of the job."""
        try:
            keys = ("audio", "poses", "clip_index")
            host = [{k: (b[k].cpu().pin_memory() if k in keys else b[k]) for k in b} for b in batches]
            for hb in host:
                hb["speaker_stat"] = {k: v.cpu().pin_memory() for k, v in hb["speaker_stat"].items()}
            nbytes = sum(host[0][k].numel() * host[0][k].element_size() for k in keys) + sum(v.numel() * v.element_size() for v in host[0]["speaker_stat"].values())
            copy_stream = torch.cuda.Stream(device=dev)

            def to_dev(hb):
                d = {k: (hb[k].to(dev, non_blocking=True) if k in keys else hb[k]) for k in hb}
                d["speaker_stat"] = {k: v.to(dev, non_blocking=True) for k, v in hb["speaker_stat"].items()}
                return d

            def run(prefetch, n):
                cur_stream = torch.cuda.current_stream(dev)
                nxt = None
                if prefetch:
                    with torch.cuda.stream(copy_stream):
                        nxt = to_dev(host[0])
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                sync()
                t0p = time.perf_counter()
                for i in range(n):
                    if prefetch:
                        cur_stream.wait_event(ev)
                        b = nxt
                        for t in list(b[k] for k in keys) + list(b["speaker_stat"].values()):
                            t.record_stream(cur_stream)
                        with torch.cuda.stream(copy_stream):
                            nxt = to_dev(host[(i + 1) % len(host)])
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                    else:
                        b = to_dev(host[i % len(host)])
                    lo, _ = pipe.forward_backward(b)
                    pipe.optimizer_updates(lo)
                sync()
                return (time.perf_counter() - t0p) * 1e3 / n

            run(True, 3)
            n = 20
            ms_pref, ms_ser = run(True, n), run(False, n)
            return {"unit": "clips/s", "steps": n, "mb_per_batch": nbytes / 1e6,
                    "prefetched": {"value": B / (ms_pref * 1e-3), "ms_per_step": ms_pref}, "serial": {"value": B / (ms_ser * 1e-3), "ms_per_step": ms_ser},
                    "note": "the timed step fed from PINNED HOST batches (the reference's boundary: collated CPU tensors moved with .cuda()); prefetched = "
                            "next batch copied on a copy stream under the current step, serial = copies in front of the step on its own stream; never `value`"}
        except Exception as e:  # informational: must not take the bench line with it
            return {"error": repr(e)[:300]}

    pcie = None
    if not stub and on_gpu and world == 1 and not forced_dp and not args.graph and not args.no_alt_mode and args.storage == "f32" and args.conv_math == "f32":
        pcie = pcie_leg()
    dp_graph = None
    legs_ok = world == 1 or args.dp_legs  # (a forced 1-rank group keeps them: that is where they are tested)
    if not stub and on_gpu and (world > 1 or forced_dp) and legs_ok and not args.graph and args.config == "voice2pose_sdt_bp" and not args.no_alt_mode:
        dp_graph = graph_leg_f32()  # every rank (collectives inside)
    alt = None
    if not stub and not args.no_alt_mode and legs_ok and args.conv_math == "f32" and args.storage == "f32" and not args.graph and on_gpu \
            and args.config == "voice2pose_sdt_bp":
        try:
            alt = bf16_leg()  # every rank (collectives inside)
        except Exception as e:  # (as graph_leg_f32: informational, must not take the bench line with it)
            if world == 1 and not forced_dp:
                raise
            alt = {"mode": "bf16", "error": repr(e)[:300]}

    if rank == 0:
        out = {
            "metric": "training clips/sec (64-frame, 137-kpt) %s" % args.config,  # BASELINE.json's metric is quoted on the default config
            "value": world * B * args.steps / elapsed, "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("bf16 (Conv2d chain: bf16 tensors in HBM, bf16 MFMA products, fp32 accumulation / statistics / master weights / gradients; "
                      "generator Conv1d chain: fp32 tensors, bf16 MFMA products; 1-D weight gradients, head, losses fp32)" if args.storage == "bf16" else
                      (("f32 (Conv2d forward / input gradient: each fp32 operand split exactly into three bf16 numbers, six bf16 MFMA products per fp32 product, "
                        "fp32 accumulation -- products exact to 2^-23; everything else fp32 arithmetic)" if (ops is not None and getattr(ops, "F32_SPLIT", False)) else "f32")
                       if args.conv_math == "f32" else "f32 storage/accumulate, %s conv products (fwd+dX)" % args.conv_math)),
            "data": "synthetic",
            "config": {"workload": "%s: %d clips/GPU x %d GPU, 64 frames, 121-kpt network I/O (137-kpt clips), L=68266 audio, "
                                   "N=%d clip codes; full train_step (mel+G fwd/bwd+L1+KL+pose-encoder x2+f64 metrics+Adam)"
                                   % (args.config, B, world, N_CLIPS),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "graph": bool(args.graph), "deterministic_dw": not args.atomic_dw, "streamk_reserved_slots": int(getattr(ops, "SK_RESERVED_SLOTS", 0)) if ops is not None else 0,
                       # run-to-run bit-identical weights in this mode (tests/test_model_gpu.py::test_train_steps_repeat_bit_identically):
                       # ordered weight-gradient / bias reductions; the normalisation statistics are fp64 atomics whose rounding to
                       # fp32 depends on the arrival order with probability ~2e-6 per step (fp64 sums of fp32 partials are exact; docs/DESIGN_rounds_1-5.md section 2)
                       "deterministic": bool(not args.atomic_dw and not args.no_streamk_dw and args.conv_math == "f32")},
            "final_G_loss": final_loss,
            "streamk_errors": 0 if (ops is not None and on_gpu and not stub) else None,  # asserted above: no stream-K launch lost a partner
            # SURVEY.md 8d's definition: clips / MEDIAN step time (per-step events on the main stream, max over ranks); `value`
            # above stays the driver's wall-clock mean over all K steps, the event-instrumented ones included
            "median_ms_per_step": median_ms, "value_at_median": world * B / (median_ms * 1e-3),
            "event_instrumented_steps": sampled, "step_ms": [round(t, 3) for t in step_ms],
            "ms_per_step_uninstrumented": clean_mean_ms, "value_uninstrumented": world * B / (clean_mean_ms * 1e-3),
        }
        if stub:
            out["stub"] = True
        if per_rank is not None:
            out["per_rank"] = per_rank
        if prof is not None and prof_steps > 0:
            summ = prof.summary()
            name, d = max(summ.items(), key=lambda kv: kv[1]["us"])
            n_dom_used = 0
            if n_dom > 0 and prof_dom.only == {name}:  # merge the dominant kernel's launches of the dominant-only "alone" steps
                dd = prof_dom.summary().get(name)
                if dd is not None:
                    d = dict(d, roles={k: list(v) for k, v in d["roles"].items()})
                    for k in ("launches", "us", "flops", "bytes"):
                        d[k] += dd[k]
                    d["max_launch_tflops"] = max(d["max_launch_tflops"], dd["max_launch_tflops"])
                    for r, v in dd["roles"].items():
                        acc = d["roles"].setdefault(r, [0, 0.0, 0.0])
                        acc[0], acc[1], acc[2] = acc[0] + v[0], acc[1] + v[1], acc[2] + v[2]
                    n_dom_used = n_dom
            dom_steps = prof_steps + n_dom_used  # sampled steps behind the dominant kernel's figures
            avg_us = d["us"] / d["launches"]
            flops_per_launch = d["flops"] / d["launches"]
            achieved = flops_per_launch / (avg_us * 1e-6) / 1e12
            # accounting guard: an event window contains the launch, so algorithmic FLOPs / window can never exceed
            # the matrix peak -- if it does, the FLOP numerator counts work that does not exist (e.g. culled structural zeros)
            for kname, kd in summ.items():
                assert kd["max_launch_tflops"] <= matrix_peak_tflops(kname) or args.conv_math != "f32", \
                    "%s: a launch 'achieved' %.1f TFLOP/s > its matrix peak: FLOP accounting is wrong" % (kname, kd["max_launch_tflops"])
            peak = matrix_peak_tflops(name)
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak,
                               "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                               "peak_is": ("dense bf16 MFMA peak (%.0f TFLOP/s) / 6: the split-fp32 kernel spends six bf16 MFMA products on every fp32 product"
                                           % BF16_MATRIX_PEAK_TFLOPS) if peak != FP32_MATRIX_PEAK_TFLOPS else "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
                               "launches_per_step": d["launches"] / dom_steps, "avg_launch_us": avg_us,
                               "event_sampled_steps": dom_steps, "event_sampled_steps_all_kernels": prof_steps,
                               "measured": "HIP events on the launching stream, sampled steps of the timed region with the "
                                           "weight-gradient side stream switched off (the kernel alone on the GPU)",
                               "algorithmic_gflop_per_launch": flops_per_launch / 1e9,
                               "algorithmic_mb_per_launch": d["bytes"] / d["launches"] / 1e6,
                               "fastest_launch_tflops": d["max_launch_tflops"]}
            # the same kernel serves the MFMA-bound Conv2d launches and the latency-bound 1-D launches (M = B*T <= 2048 rows):
            # the average above mixes them, the split shows each (role = forward / input gradient, 2-D / 1-D stage)
            out["roofline"]["by_role"] = {
                r: {"launches_per_step": v[0] / dom_steps, "avg_launch_us": v[1] / v[0], "achieved": v[2] / (v[1] * 1e-6) / 1e12,
                    "frac": v[2] / (v[1] * 1e-6) / 1e12 / peak} for r, v in sorted(d["roles"].items())}
            out["roofline"].update(cited_traffic(name))
            out["roofline"].update(cited_sustained_mhz())
            # the whole step against the fp32-MFMA floor of its convolutions: algorithmic conv FLOPs of a step / matrix peak / step time
            step_gflop = sum(v["flops"] for v in summ.values()) / prof_steps / 1e9
            out["roofline"]["step_gflop"] = step_gflop
            # ONE peak convention per object: step_frac is quoted against the SAME peak as frac (the dominant kernel's); the figure
            # against the fp32-MFMA floor of rounds 1-4 keeps its own, explicit name
            out["roofline"]["step_frac"] = step_gflop * 1e9 / (peak * 1e12) / (clean_mean_ms * 1e-3)
            out["roofline"]["step_frac_vs_fp32_mfma"] = step_gflop * 1e9 / (FP32_MATRIX_PEAK_TFLOPS * 1e12) / (clean_mean_ms * 1e-3)
            out["roofline"]["peak_clock_mhz"] = 2400  # every peak in this line is at the 2.4 GHz boost clock; sustained_mhz is what the PMC run saw
            if out["roofline"].get("traffic"):
                out["roofline"]["traffic_over_algorithmic"] = out["roofline"]["traffic"] / (d["bytes"] / d["launches"])
            out["conv_kernels_peak_tflops"] = {k: matrix_peak_tflops(k) for k in sorted(summ)}
            out["roofline"]["trace_based"] = cited_trace_fraction(name)
            if prof_ovl is not None and n_ovl > 0:
                do = prof_ovl.summary()[name]
                out["roofline"]["overlapped"] = {
                    "avg_launch_us": do["us"] / do["launches"], "achieved": do["flops"] / (do["us"] * 1e-6) / 1e12,
                    "event_sampled_steps": n_ovl,
                    "note": "as run by default: weight-gradient kernels execute concurrently on a second stream, so a launch's "
                            "duration includes the time it shared the GPU (this is what a kernel trace of this command shows)"}
            tot_us = sum(v["us"] for v in summ.values())
            tot_fl = sum(v["flops"] for v in summ.values())
            out["conv_kernels"] = {k: {"launches_per_step": v["launches"] / prof_steps, "ms_per_step": v["us"] / prof_steps / 1e3,
                                       "tflops": v["flops"] / (v["us"] * 1e-6) / 1e12} for k, v in sorted(summ.items())}
            out["conv_total"] = {"ms_per_step": tot_us / prof_steps / 1e3, "tflops": tot_fl / (tot_us * 1e-6) / 1e12,
                                 "gflop_per_step": tot_fl / prof_steps / 1e9}
            if stages is not None and n_stage > 0 and args.config.startswith("voice2pose"):
                win = {k: v[1] / v[0] for k, v in stages.windows_us().items()}  # us per step, per window
                if all(k in win for k in ("g1d_fwd", "g1d_bwd")):
                    scale = B / 32.0
                    mb = sum(CONV1D_MB_PER_32.values()) * scale  # weights are counted once per step; activations scale with B
                    gflop = sum(CONV1D_GFLOP_PER_CLIP.values()) * B
                    stage_us = sum(win.values())
                    exposed_us = win["g1d_fwd"] + win["g1d_bwd"]
                    tbs = mb * 1e6 / (stage_us * 1e-6) / 1e12
                    out["roofline_conv1d"] = {
                        "bound": "hbm", "achieved": tbs * 1e3, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": tbs / HBM_PEAK_TBS,
                        "mfma_achieved_tflops": gflop * 1e9 / (stage_us * 1e-6) / 1e12,
                        "mfma_frac": gflop * 1e9 / (stage_us * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
                        "algorithmic_mb_per_step": mb, "algorithmic_gflop_per_step": gflop,
                        "stage_us_per_step": stage_us, "windows_us": win, "event_sampled_steps": n_stage, "exposed_us_per_step": exposed_us,
                        "note": "Conv1d stacks of one train step: U-Net + decoder forward and backward chains (main stream, exposed), "
                                "their weight gradients and the two no-grad pose-encoder passes (side stream, overlapped with the "
                                "Conv2d backward); HIP-event windows on the stream each piece runs on, sampled steps that carry ONLY these 8 events; "
                                "algorithmic bytes / FLOPs from SURVEY.md 8d.  g1d_fwd / g1d_bwd: the generator's sixteen blocks as ONE persistent "
                                "launch per direction (csrc/chain1d.hip; --no-chain1d: two to three launches per block) + resize / head / loss "
                                "kernels.  In fp32 the stage is bound by the fp32 MFMA rate and by hand-off / launch latency, not by HBM: at the "
                                "MFMA roofline (%.0f us) it would still reach only %.0f %% of 8 TB/s"
                                % (gflop * 1e9 / (FP32_MATRIX_PEAK_TFLOPS * 1e12) * 1e6,
                                   100 * mb * 1e6 / (gflop * 1e9 / (FP32_MATRIX_PEAK_TFLOPS * 1e12)) / 1e12 / HBM_PEAK_TBS)}
            hbs = sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_hbm_kernels.txt")) if os.path.isdir(os.path.join(REPO, "profiles")) else []
            if hbs:
                out["hbm_kernels_table"] = "profiles/%s (tools/hbm_kernels.py over a rocprofv3 kernel trace of this command)" % hbs[-1]
        if dp_graph is not None:
            if "value" in dp_graph:
                dp_graph["vs_default"] = dp_graph["value"] / out["value_uninstrumented"]
            out["dp_graph_replay"] = dp_graph
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if alt is not None:
            if "value" in alt:
                alt["vs_default"] = alt["value"] / out["value_uninstrumented"]
            out["alt_conv_math"] = alt
        if world == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline(B)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1 or forced_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
