#!/usr/bin/env python
"""Benchmark of the SDT voice2pose training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): voice2pose_sdt_bp, 32 clips per GPU (weak scaling), 64-frame / 121-kpt clips
(137 on-disk keypoints), fp32, synthetic seeded clips pre-staged in HBM.  One step = Voice2Pose.train_step without
logging: mel -> generator -> L1 + clip-code KL -> no-grad pose encoder x2 -> float64 metrics -> backward ->
gradient all-reduce (N>1) -> Adam on the clip-code table and the generator.  Prints ONE JSON line on rank 0.

The JSON carries `roofline` for the dominant kernel (the fp32-MFMA implicit-GEMM conv instantiation with the most
time): algorithmic FLOPs per launch / mean launch duration from HIP events recorded over the timed region, against
the 157.3 TFLOP/s fp32 matrix peak; and `cpu_baseline`: the CPU oracle (PyTorch-CPU restatement of the reference
step, oracle/) timed on this box's host cores for a few steps of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
N_CLIPS = 4096
# timed steps that carry HIP events around every conv launch, half of them "alone", half "as run" (see below).  A sampled
# step costs ~1.6 ms extra (the events serialise neighbouring launches, and an "alone" step gives up the side stream), so they
# are kept to ~1 in 15: 2 of the default 30 steps, 4 from 60 steps on
EVENT_STEPS_MAX = int(os.environ.get("SDT_EVENT_STEPS", "4"))


def event_steps(steps):
    return min(EVENT_STEPS_MAX, max(2, (steps // 30) * 2), steps)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--config", default="voice2pose_sdt_bp")
    ap.add_argument("--conv-math", default="f32", choices=["f32", "bf16x6", "bf16x3", "bf16"],
                    help="product arithmetic of the forward / input-gradient conv kernels (experiments; the metric is quoted on f32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-launch HIP events (roofline leg)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (N=1)")
    ap.add_argument("--graph-streams", action="store_true", help="with --graph: capture the side streams too (experiment)")
    ap.add_argument("--no-overlap-aux", action="store_true", help="keep the no-grad pose-encoder passes on the main stream")
    ap.add_argument("--no-stats-fusion", action="store_true", help="separate statistics pass for the 2-D norms (A/B of the fused conv epilogue)")
    ap.add_argument("--no-overlap-dw", action="store_true", help="keep the weight-gradient kernels on the main stream (default: side stream, +4 %)")
    ap.add_argument("--no-defer-dw", action="store_true", help="launch the 1-D stage's weight gradients inline (default: one batch on the side stream under the Conv2d backward)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        try:  # "nccl" is RCCL on ROCm; device_id binds the communicator to this rank's GPU up front
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        except TypeError:
            dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    dev = torch.device("cuda", local_rank)

    from __graft_entry__ import make_pipeline
    from speechdrivestemplates_amd import ops
    B = args.batch
    ops.OVERLAP_DW = not args.no_overlap_dw
    ops.DEFER_SMALL_DW = not args.no_defer_dw
    ops.PROFILER_NO_FUSION = bool(args.no_stats_fusion)
    ops.CAPTURE_SIDE_STREAMS = bool(args.graph_streams)
    ops.OVERLAP_AUX = not args.no_overlap_aux
    ops.set_conv_math(args.conv_math)
    pipe, cfg = make_pipeline(args.config, N_CLIPS, batch_global=B * world)
    batches = stage_batches(4, B, rank, dev)

    def step(i):
        losses, _ = pipe.forward_backward(batches[i % len(batches)])
        pipe.optimizer_updates(losses)
        return losses

    runner = step
    if args.graph and world == 1:
        from speechdrivestemplates_amd.graph import GraphedStep
        gs = GraphedStep(pipe, warmup=min(3, max(1, args.warmup - 1)))
        runner = lambda i: gs.run(batches[i % len(batches)])  # noqa: E731

    for i in range(args.warmup):
        runner(i)
    # HIP events around every conv launch on a few timed steps, recorded on the stream each kernel is launched on.  By default the
    # weight-gradient kernels run on a second stream, concurrently with the input-gradient chain, so a launch's duration
    # then includes the time it shared the GPU: sampled steps therefore ALTERNATE between "alone" (side stream off for
    # that step: the kernel-quality figure reported as roofline.achieved) and "as run" (roofline.overlapped).
    prof = prof_ovl = None
    if not args.no_kernel_events and not (args.graph and world == 1):
        prof = ops.ConvProfiler(pool=2 * 200 * EVENT_STEPS_MAX)
        prof_ovl = ops.ConvProfiler(pool=2 * 200 * EVENT_STEPS_MAX) if ops.OVERLAP_DW else None
    n_ev = event_steps(args.steps) if prof is not None else 0
    sampled = sorted({(j + 1) * args.steps // (n_ev + 1) for j in range(n_ev)}) if n_ev else []  # spread over the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    overlap_dw = ops.OVERLAP_DW
    n_alone = n_ovl = 0
    for i in range(args.steps):
        if prof is not None:  # sampled: the events serialise the host a little and cost a few % on the steps they cover
            ops.PROFILER, ops.OVERLAP_DW = None, overlap_dw
            if i in sampled:
                if prof_ovl is not None and n_alone > n_ovl:
                    ops.PROFILER, n_ovl = prof_ovl, n_ovl + 1
                else:
                    ops.PROFILER, ops.OVERLAP_DW, n_alone = prof, False, n_alone + 1
        losses = runner(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.PROFILER, ops.OVERLAP_DW = None, overlap_dw
    prof_steps = n_alone
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(losses["G_loss" if "G_loss" in losses else "loss"].detach())
    assert final_loss == final_loss and final_loss < 10.0, "training diverged: G_loss=%r" % final_loss

    if rank == 0:
        out = {
            "metric": "training clips/sec (64-frame, 137-kpt) voice2pose_sdt_bp",
            "value": world * B * args.steps / elapsed, "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.conv_math == "f32" else "f32 storage/accumulate, %s conv products (fwd+dX)" % args.conv_math,
            "data": "synthetic",
            "config": {"workload": "%s: %d clips/GPU x %d GPU, 64 frames, 121-kpt network I/O (137-kpt clips), L=68266 audio, "
                                   "N=%d clip codes; full train_step (mel+G fwd/bwd+L1+KL+pose-encoder x2+f64 metrics+Adam)"
                                   % (args.config, B, world, N_CLIPS),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "graph": bool(args.graph and world == 1)},
            "final_G_loss": final_loss,
        }
        if prof is not None and prof_steps > 0:
            summ = prof.summary()
            name, d = max(summ.items(), key=lambda kv: kv[1]["us"])
            avg_us = d["us"] / d["launches"]
            flops_per_launch = d["flops"] / d["launches"]
            achieved = flops_per_launch / (avg_us * 1e-6) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": None,
                               "launches_per_step": d["launches"] / prof_steps, "avg_launch_us": avg_us,
                               "event_sampled_steps": prof_steps,
                               "measured": "HIP events on the launching stream, sampled steps of the timed region with the "
                                           "weight-gradient side stream switched off (the kernel alone on the GPU)",
                               "algorithmic_gflop_per_launch": flops_per_launch / 1e9,
                               "algorithmic_mb_per_launch": d["bytes"] / d["launches"] / 1e6}
            # HBM-side traffic cannot be read from inside the process: cite the committed rocprofv3 PMC measurement of this
            # same command (two separate --pmc passes, gfx950 x2 correction on FETCH_SIZE; profiles/README.md)
            tpath = os.path.join(REPO, "profiles", "r01_hbm_traffic_bench.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("dominant_kernel", "") == name:
                    out["roofline"]["traffic"] = tj["traffic_mb_per_launch"] * 1e6
                    out["roofline"]["traffic_unit"] = "bytes/launch (2*FETCH_SIZE+WRITE_SIZE, fabric side, upper bound on HBM)"
                    out["roofline"]["traffic_source"] = "profiles/r01_hbm_traffic_bench.json"
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
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
