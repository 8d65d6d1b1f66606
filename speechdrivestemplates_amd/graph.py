"""hipGraph replay of a whole training step.

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

    def _eager(self, batch):
        losses, _ = self.pipe.forward_backward(batch)
        self.pipe.optimizer_updates(losses)
        return losses

    @staticmethod
    def _clone_batch(batch):
        def cp(v):
            if torch.is_tensor(v):
                return v.clone()
            if isinstance(v, dict):
                return {k: cp(x) for k, x in v.items()}
            return v
        return {k: cp(v) for k, v in batch.items()}

    @staticmethod
    def _copy_into(dst, src):
        for k, v in src.items():
            if torch.is_tensor(v):
                if v.is_cuda:
                    dst[k].copy_(v, non_blocking=True)
            elif isinstance(v, dict):
                GraphedStep._copy_into(dst[k], v)

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
