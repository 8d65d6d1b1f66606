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
    q.put((rank, worst, touched, float(scal["rank"]), float(scal["G_loss"]), float(losses["G_loss"])))
    dist.destroy_process_group()


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
    res = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, worst0, touched0, mean_rank, mean_loss, loss0), (r1, worst1, touched1, _, _, loss1) = res
    # fp32 reduction-order noise only (early-layer weight gradients are noisy in fp32)
    assert worst0 < 5e-3 and worst1 < 5e-3, (worst0, worst1)
    assert touched0 == touched1 == [0, 1, 2, 3], (touched0, touched1)  # rows of BOTH ranks, present on both
    assert mean_rank == 0.5  # rank 0 holds the mean over ranks after the packed reduce
    assert abs(mean_loss - 0.5 * (loss0 + loss1)) < 1e-6
