"""hipGraph replay of a whole training step (``SYS.HIP_GRAPH: True``; product since round 4, data-parallel since round 5).

Why: the bf16-storage step needs ~2-3 ms of GPU time but the host needs 3-5 ms to enqueue its ~270 launches one by one (pose2pose: 1.2 ms of GPU
work behind 3.9 ms of enqueueing) -- replayed from a graph the host cost is a handful of device-to-device input copies plus one graph launch.
The fp32 voice2pose step is GPU-bound and gains nothing.

``GraphedStep`` captures ``forward_backward`` + ``optimizer_updates`` of a pipeline (every HIP kernel, memset and torch glue op on the step's
stream, autograd's backward and the Adam kernels included).  Everything the step needs is graph-safe by construction: no host synchronisation
(the KL "skip" predicate, the Adam step counter and the learning rate live on the device), static shapes, workspace allocations from the graph's
private pool, persistent-launch flags / counters that their consumers lower again (a replay starts from the same zeroed words).

Data parallelism (round 5; the reference's DDP configs 4 and 5, main.py:53-67) -- two forms, chosen by the process group's backend:
  * ``full``  (backend "nccl" = RCCL): the gradient all-reduces are captured WITH the step -- the bucket launches from the backward hooks on the
    communication stream (a fork of the capturing stream), the late buckets and the join before Adam.  ProcessGroupNCCL enqueues its kernels on
    its own stream behind an event of the launching stream; inside a capture those become graph edges, so the replayed graph keeps the overlap
    of exchange and backward that the eager step has.
  * ``split`` (any other backend, e.g. gloo in the tests; or SDT_GRAPH_DP=split): collectives that cannot be captured cut the step into graph
    SEGMENTS with the exchange issued eagerly between them: [forward + backward] -> all-reduce -> [Adam (+ the discriminator's backward)] ->
    all-reduce -> [Adam of the discriminator].  The bucket hooks are muted during such a capture (one exchange per optimiser group after the
    backward pass: no overlap, but still no per-launch host cost).  Segments share one memory pool and are replayed in capture order.
Either way a replayed step computes exactly what the eager data-parallel step computes (tests/test_dp_gpu.py).
"""
import os

import torch


class GraphedStep:
    def __init__(self, pipe, warmup=3, mode=None):
        self.pipe = pipe
        self.warmup = warmup
        self.calls = 0
        self.segments = None  # [("graph", CUDAGraph) | ("eager", callable)] in replay order
        self.static = None
        self.losses = None
        self.results = None  # results dict of the captured step: static tensors that every replay overwrites
        reducer = getattr(pipe, "reducer", None)
        if mode is None:
            if reducer is None or not reducer.active:
                mode = "single"
            else:
                import torch.distributed as dist
                mode = os.environ.get("SDT_GRAPH_DP") or ("full" if dist.get_backend() == "nccl" else "split")
        assert mode in ("single", "full", "split"), mode
        self.mode = mode

    @property
    def graph(self):  # (round-4 attribute: the first captured graph)
        return None if not self.segments else next((g for kind, g in self.segments if kind == "graph"), None)

    def _eager(self, batch):
        losses, results = self.pipe.forward_backward(batch)
        self.pipe.optimizer_updates(losses)
        self.results = results
        return losses

    @staticmethod
    def _clone_batch(batch):
        """Static device copies of every tensor field (``num_frames`` stays where it is: it is read on the host)."""
        dev = torch.device('cuda', torch.cuda.current_device())

        def cp(k, v):
            if torch.is_tensor(v):
                return v.clone() if (k == 'num_frames' or v.is_cuda) else v.to(dev)
            if isinstance(v, dict):
                return {kk: cp(kk, x) for kk, x in v.items()}
            return v
        return {k: cp(k, v) for k, v in batch.items()}

    @staticmethod
    def _copy_into(dst, src):
        """Refresh the captured graph's static inputs.  Host tensors are copied H2D as well (a DataLoader batch); the one field
        the step reads on the HOST at capture time, ``num_frames``, and every shape must equal the captured ones -- a replay
        cannot follow them, so a mismatch raises instead of silently replaying the captured batch."""
        for k, v in src.items():
            if torch.is_tensor(v):
                if v.shape != dst[k].shape or v.dtype != dst[k].dtype:
                    raise RuntimeError('GraphedStep: field %r changed shape/dtype since capture (%s %s -> %s %s)'
                                       % (k, tuple(dst[k].shape), dst[k].dtype, tuple(v.shape), v.dtype))
                if k == 'num_frames':
                    if not torch.equal(v.cpu(), dst[k].cpu()):
                        raise RuntimeError('GraphedStep: num_frames differs from the captured value')
                    continue
                dst[k].copy_(v, non_blocking=True)
            elif isinstance(v, dict):
                GraphedStep._copy_into(dst[k], v)
            elif k == 'speaker':
                # the step resolves speaker[0]'s normalisation statistics on the HOST at capture time and bakes them into the graph:
                # a batch of another speaker must not replay with them
                if list(v)[:1] != list(dst[k])[:1]:
                    raise RuntimeError('GraphedStep: speaker %r differs from the captured %r (its statistics are part of the graph)'
                                       % (list(v)[:1], list(dst[k])[:1]))
            elif not GraphedStep._same(v, dst[k]):
                raise RuntimeError('GraphedStep: non-tensor field %r differs from the captured value' % k)

    @staticmethod
    def _same(a, b):
        """equality that is safe for numpy arrays / lists of arrays (``a != b`` on arrays has no truth value)"""
        import numpy as np
        if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
            return isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.shape == b.shape and bool(np.array_equal(a, b))
        if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
            return len(a) == len(b) and all(GraphedStep._same(x, y) for x, y in zip(a, b))
        return a == b

    # -- capture ---------------------------------------------------------------------------------------
    def _capture(self):
        """Run the step once under stream capture.  ``single`` / ``full``: one graph.  ``split``: the reducer calls ``_cut`` wherever the step
        exchanges gradients; the capture is closed there, the exchange recorded as an eager item, and a new segment opened on the same pool."""
        torch.cuda.synchronize()
        segments = []
        reducer = getattr(self.pipe, "reducer", None)
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        from . import ops
        ops.prepare_capture_stream(torch.device('cuda', torch.cuda.current_device()), stream)  # error words must survive replays
        state = {"g": None, "pool": torch.cuda.graph_pool_handle()}  # one private pool for all segments: a tensor allocated in one is read in the next

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread_local: ProcessGroupNCCL's watchdog thread polls the events of earlier (eager) collectives with hipEventQuery; under the default
            # global capture mode such a call from ANY thread fails ("operation not permitted when stream is capturing") and invalidates the capture
            # -- seen as an intermittent abort of the data-parallel capture (one run in two).  Only this thread's calls need to be capture-safe.
            g.capture_begin(pool=state["pool"], capture_error_mode="thread_local")
            state["g"] = g

        def end():
            state["g"].capture_end()
            segments.append(("graph", state["g"]))
            state["g"] = None

        def cut(exchange):
            end()
            segments.append(("eager", exchange))  # not executed now: nothing of this step has run yet
            begin()

        if self.mode == "split":
            reducer.capture_cut = cut
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        try:
            with torch.cuda.stream(stream):
                begin()
                try:
                    self.losses = self._eager(self.static)  # recorded, not executed
                finally:
                    if state["g"] is not None:
                        end()
        finally:
            if self.mode == "split":
                reducer.capture_cut = None
        torch.cuda.current_stream().wait_stream(stream)
        self.segments = segments
        self._stream = stream

    def run(self, batch):
        """One training step on ``batch`` (device tensors with the collated layout).  The first ``warmup`` calls run
        eagerly (they are real steps); the next call captures the graph(s), and every call from then on replays."""
        self.calls += 1
        if self.calls <= self.warmup:
            return self._eager(batch)
        if self.segments is None:
            self.static = self._clone_batch(batch)
            self._capture()
        else:
            self._copy_into(self.static, batch)
        for kind, item in self.segments:
            if kind == "graph":
                item.replay()
            else:
                item()
        return self.losses
