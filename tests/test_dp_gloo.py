"""Data-parallel path on CPU: two gloo ranks run dp.GradReducer / dp.reduce_scalars over flat gradient buffers
whose contents come from the CPU oracle (the HIP kernels cannot run here).  Checks the DP contract of DESIGN.md:
  * after the summing all-reduce and the 1/world scale folded into Adam, every rank holds the same gradients, equal
    to the single-process gradients of the concatenated batch for the per-sample-normalised sdt generator
    (IN2d / channel-LN have no cross-sample statistics -> data parallelism is exact for the L1 term);
  * clip-code rows touched by different ranks are disjoint and end up averaged (DDP semantics);
  * the packed scalar reduce gives rank 0 the mean of every loss in one collective."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


class _FlatGroup:
    """What dp.GradReducer needs from an optimiser group: a flat gradient buffer and a grad_scale slot."""

    def __init__(self, tensors):
        self.tensors = tensors
        self.flat_grad = torch.cat([t.grad.reshape(-1) for t in tensors])
        self.grad_scale = 1.0

    def scatter_back(self):
        off = 0
        out = []
        for t in self.tensors:
            out.append(self.flat_grad[off:off + t.numel()].view_as(t) * self.grad_scale)
            off += t.numel()
        return out


def _grads(cfg, state, batch, O):
    for v in state.values():
        if v.is_floating_point():
            v.grad = None
    losses, _ = O.voice2pose_forward(state, batch, cfg, True)
    losses["G_loss"].backward()
    return losses


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import sdt_oracle as O
    from speechdrivestemplates_amd import dp
    cfg = O.cfg_named("voice2pose_sdt_bp")
    cfg.VOICE2POSE.GENERATOR.LAMBDA_CLIP_KL = 0.0  # the KL term uses per-rank batch statistics by design (reference semantics)
    n_clips, B = 8, 2
    state = O.make_voice2pose_state(cfg, n_clips, seed=0, code_std=0.5)
    O.OracleVoice2Pose(cfg, state)  # marks the trainable leaves
    full = O.make_batch(B * world, n_clips, step=0, seed=1)
    mine = {k: (v[rank * B:(rank + 1) * B] if torch.is_tensor(v) else v) for k, v in full.items() if k != "speaker_stat"}
    mine["num_frames"] = full["num_frames"][:B]
    losses = _grads(cfg, state, mine, O)
    params = [v for k, v in state.items() if v.requires_grad and (k.startswith("netG.") or k == "clips_code")]
    grp = _FlatGroup(params)
    red = dp.GradReducer([grp], overlap=False)
    assert red.ws == world and grp.grad_scale == 1.0 / world
    red.all_reduce()
    averaged = grp.scatter_back()
    # reference: one process, whole batch
    ref_state = O.make_voice2pose_state(cfg, n_clips, seed=0, code_std=0.5)
    O.OracleVoice2Pose(cfg, ref_state)
    _grads(cfg, ref_state, full, O)
    ref = [v.grad for k, v in ref_state.items() if v.requires_grad and (k.startswith("netG.") or k == "clips_code")]
    worst = 0.0
    for a, r in zip(averaged, ref):
        worst = max(worst, ((a - r).abs().max() / r.abs().max().clamp_min(1e-12)).item())
    # clip-code rows: each rank touched its own rows only; after the exchange both ranks see all of them, averaged
    code_avg = averaged[[k for k, v in state.items() if v.requires_grad and (k.startswith("netG.") or k == "clips_code")].index("clips_code")]
    touched = sorted(set(code_avg.abs().sum(1).nonzero().flatten().tolist()))
    scal = dp.reduce_scalars({"G_loss": losses["G_loss"].detach(), "rank": torch.tensor(float(rank))})
    # the kernel error flag rides in the same message: every rank learns how many ranks raised it (here: rank 1 only), the scalars are unchanged
    d2 = {"G_loss": losses["G_loss"].detach().clone(), "rank": torch.tensor(float(rank))}
    n_bad = dp.reduce_scalars(d2, error_flag=torch.tensor(1.0 if rank == 1 else 0.0, dtype=torch.float64))
    assert n_bad == 1.0 and dp.any_rank_flag(torch.tensor(0.0, dtype=torch.float64)) == 0.0
    assert dp.any_rank_flag(torch.tensor(3.0 if rank == 0 else 0.0, dtype=torch.float64)) == 1.0
    if rank == 0:
        assert float(d2["rank"]) == float(scal["rank"]) and abs(float(d2["G_loss"]) - float(scal["G_loss"])) < 1e-12
    q.put((rank, worst, touched, float(scal["rank"]), float(scal["G_loss"]), float(losses["G_loss"])))
    dist.destroy_process_group()



def _collect(procs, q, n, timeout):
    """n results from the worker queue; fails fast (instead of waiting out the timeout) when a worker has died"""
    import queue as _queue
    import time as _time
    out, t0 = [], _time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=5))
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, "worker exited with %s" % dead
            assert _time.time() - t0 < timeout, "timed out waiting for the workers"
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_rank_gradient_exchange_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, len(procs), 500))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, worst0, touched0, mean_rank, mean_loss, loss0), (r1, worst1, touched1, _, _, loss1) = res
    # fp32 reduction-order noise only (early-layer weight gradients are noisy in fp32)
    assert worst0 < 5e-3 and worst1 < 5e-3, (worst0, worst1)
    assert touched0 == touched1 == [0, 1, 2, 3], (touched0, touched1)  # rows of BOTH ranks, present on both
    assert mean_rank == 0.5  # rank 0 holds the mean over ranks after the packed reduce
    assert abs(mean_loss - 0.5 * (loss0 + loss1)) < 1e-6


# ------------------------------------------------------------------------------------------------------------------
# Replica synchronisation (DDP-constructor semantics, reference voice2pose.py:222-223) with DIFFERENT per-rank seeds
# ------------------------------------------------------------------------------------------------------------------
class _CpuFlatAdam:
    """CPU stand-in with optim.FlatAdam's attribute surface (the real one launches a HIP kernel): parameters re-homed
    into one flat buffer in the kernels' (Cout,k,Cin) memory order, torch's Adam rule on the flat views."""

    class _Mirrors:
        dirty = False

        def mark_dirty(self):
            self.dirty = True

    def __init__(self, params, lr):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        self.flat_param, self.flat_grad = torch.zeros(n), torch.zeros(n)
        self.exp_avg, self.exp_avg_sq = torch.zeros(n), torch.zeros(n)
        self.state_dev, self.lr_dev = torch.zeros(2, dtype=torch.int64), torch.tensor([lr])
        self.param_groups, self._lr_host, self.grad_scale, self.mirrors = [dict(lr=lr)], lr, 1.0, self._Mirrors()
        off = 0
        self.views = []
        for p in self.params:
            perm = [0] + list(range(2, p.dim())) + [1] if p.dim() >= 3 else list(range(p.dim()))
            inv = [perm.index(d) for d in range(p.dim())]
            shape = [p.shape[d] for d in perm]
            view = self.flat_param[off:off + p.numel()].view(shape)
            view.copy_(p.data.permute(perm))
            p.data = view.permute(inv)  # a strided view, like FlatAdam's
            self.views.append((off, shape, inv))
            off += p.numel()

    def step(self):
        self.state_dev[0] += 1
        t = int(self.state_dev[0])
        g = self.flat_grad * self.grad_scale
        self.exp_avg.mul_(0.9).add_(g, alpha=0.1)
        self.exp_avg_sq.mul_(0.999).addcmul_(g, g, value=0.001)
        denom = (self.exp_avg_sq / (1 - 0.999 ** t)).sqrt_().add_(1e-8)
        self.flat_param.addcdiv_(self.exp_avg / (1 - 0.9 ** t), denom, value=-float(self.lr_dev))


def _sync_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from speechdrivestemplates_amd import dp
    torch.manual_seed(100 + rank)  # deliberately different initialisation per rank

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv1d(6, 8, 3, padding=1, bias=False)
            self.bn = torch.nn.BatchNorm1d(8)
            self.head = torch.nn.Conv1d(8, 4, 1)
            self.frozen = torch.nn.Conv1d(4, 4, 1)  # a module without an optimiser (the no-grad pose encoder's role)
            self.register_buffer("codes", torch.randn(5, 3))

        def forward(self, x):
            return self.frozen(self.head(torch.relu(self.bn(self.conv(x)))))

    net = Net()
    with torch.no_grad():
        net.bn.running_mean.normal_()
    opt = _CpuFlatAdam(list(net.conv.parameters()) + list(net.bn.parameters()) + list(net.head.parameters()), lr=1e-2 * (1 + rank))
    opt.exp_avg.normal_()  # pretend a resumed optimiser state that differs per rank
    opt.state_dev[0] = 3 + rank
    before = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    n_sent = dp.sync_replicas(net, [opt])
    assert n_sent > 0 and opt.mirrors.dirty
    # the strided parameter views still alias the flat buffer after the broadcast
    assert net.conv.weight.data_ptr() == opt.flat_param.data_ptr()
    state = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [b.detach().double().reshape(-1).float() for b in net.buffers()]
                      + [opt.exp_avg, opt.exp_avg_sq, opt.state_dev.float(), opt.lr_dev])
    # two data-parallel steps: per-rank batch, summed gradients, 1/world folded into the update
    red = dp.GradReducer([opt], overlap=False)
    x_all = torch.randn(2 * world, 6, 10, generator=torch.Generator().manual_seed(7))
    for step in range(2):
        for p in net.parameters():
            p.grad = None
        net.train()
        loss = net(x_all[rank * 2:(rank + 1) * 2] + step).pow(2).mean()
        loss.backward()
        off = 0
        for p, (o, shape, inv) in zip(opt.params, opt.views):
            perm = [inv.index(d) for d in range(p.dim())]
            opt.flat_grad[o:o + p.numel()].view(shape).copy_(p.grad.permute(perm))
        red.all_reduce()
        opt.step()
    weights = opt.flat_param.clone()
    bn_stats = net.bn.running_mean.clone()  # rank-local during training ...
    dp.sync_buffers(net)  # ... and taken from rank 0 before validation / checkpoint
    q.put((rank, before.numpy(), state.numpy(), weights.numpy(), bn_stats.numpy(), net.bn.running_mean.numpy().copy(), float(opt._lr_host)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_replicas_start_identical_despite_different_seeds_and_stay_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, len(procs), 250), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, before0, state0, w0, bn_local0, bn_synced0, lr0), (_, before1, state1, w1, bn_local1, bn_synced1, lr1) = res
    import numpy as np
    assert not np.array_equal(before0, before1)  # the seeds really differed
    assert np.array_equal(state0, state1)  # parameters (incl. the optimiser-less module), buffers, Adam moments, step, lr
    assert np.array_equal(state0[:before0.size], before0)  # ... and they are rank 0's
    assert lr0 == lr1 == pytest.approx(1e-2)
    assert np.array_equal(w0, w1)  # same averaged gradient applied to the same weights: replicas stay identical
    assert not np.array_equal(bn_local0, bn_local1)  # BatchNorm running statistics are per-rank during training (no SyncBN) ...
    assert np.array_equal(bn_synced0, bn_local0) and np.array_equal(bn_synced1, bn_local0)  # ... rank 0's win at sync points
