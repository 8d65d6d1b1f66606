"""hipGraph replay of a whole training step (product since round 4: the bf16-storage step needs 3.5 ms of GPU time, the host needs
5.6 ms to enqueue its ~270 launches one by one -- replayed from a graph the host cost is 0.16 ms per step; the fp32 step is GPU-bound
and gains nothing).  ``SYS.HIP_GRAPH: True`` makes Voice2Pose.train_step use it (single-GPU runs).

The 1-D stage of the generator is ~100 launches of a few microseconds each; issued one by one from Python they are
host-bound.  ``GraphedStep`` captures ``forward_backward`` + ``optimizer_updates`` of a Voice2Pose pipeline (every
HIP kernel, memset and torch glue op on the step's stream, including autograd's backward and the Adam kernels)
into one hipGraph and replays it per step: the host cost drops to a handful of device-to-device input copies plus
one graph launch.  Everything the step needs is already graph-safe by construction: no host synchronisation (the KL
"skip" predicate, the Adam step counter and the learning rate live on the device), static shapes, and workspace
allocations come from the graph's private pool.

Only for world_size == 1 here: with data parallelism the gradient all-reduce sits between backward and the
optimiser kernels and is issued eagerly (bench.py keeps that path un-captured).
"""
import torch


class GraphedStep:
    def __init__(self, pipe, warmup=3):
        self.pipe = pipe
        self.warmup = warmup
        self.calls = 0
        self.graph = None
        self.static = None
        self.losses = None
        self.results = None  # results dict of the captured step: static tensors that every replay overwrites

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

    def run(self, batch):
        """One training step on ``batch`` (device tensors with the collated layout).  The first ``warmup`` calls run
        eagerly (they are real steps); the next call captures the graph, and every call from then on replays it."""
        self.calls += 1
        if self.calls <= self.warmup:
            return self._eager(batch)
        if self.graph is None:
            self.static = self._clone_batch(batch)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.losses = self._eager(self.static)  # recorded, not executed
        else:
            self._copy_into(self.static, batch)
        self.graph.replay()
        return self.losses
