"""Host-side logic that does not need a GPU: the per-epoch learning-rate schedule must follow
torch.optim.lr_scheduler.MultiStepLR exactly as the reference drives it (voice2pose.py:253-279, trainer.py:172-200,396-398),
including its behaviour when training is resumed from a checkpoint."""
import pytest
import torch


class _Opt:
    """What _MultiStepLR needs of FlatAdam: param_groups[0] and sync_lr()."""

    def __init__(self, lr, group=None):
        self.param_groups = [dict(lr=lr) if group is None else dict(group)]
        self.synced = []

    def sync_lr(self):
        self.synced.append(self.param_groups[0]['lr'])


def _torch_run(E, epochs, resume_at=None):
    """lr seen during each epoch by the reference's loop; optionally save after `resume_at` epochs and resume."""
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([w], lr=1e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [E - 10, E - 2], gamma=0.1, last_epoch=-1)
    seen = []
    for epoch in range(epochs):
        if resume_at is not None and epoch == resume_at:
            sd = opt.state_dict()  # what trainer.save_checkpoint stores after `epoch` epochs
            opt = torch.optim.Adam([w], lr=1e-4)
            opt.load_state_dict(sd)
            sch = torch.optim.lr_scheduler.MultiStepLR(opt, [E - 10, E - 2], gamma=0.1, last_epoch=epoch)
        seen.append(opt.param_groups[0]['lr'])
        opt.step()
        sch.step()
    return seen


def _own_run(E, epochs, resume_at=None):
    from speechdrivestemplates_amd.core.pipelines.voice2pose import _MultiStepLR
    opt = _Opt(1e-4)
    sch = _MultiStepLR(opt, [E - 10, E - 2], 0.1, -1)
    seen = []
    for epoch in range(epochs):
        if resume_at is not None and epoch == resume_at:
            opt = _Opt(1e-4, group=opt.param_groups[0])  # FlatAdam.load_state_dict copies lr and initial_lr
            sch = _MultiStepLR(opt, [E - 10, E - 2], 0.1, epoch)
        seen.append(opt.param_groups[0]['lr'])
        sch.step()
    return seen


@pytest.mark.parametrize("E,resume_at", [(100, None), (100, 50), (100, 89), (100, 90), (100, 97), (100, 98), (12, None), (12, 1),
                                         (2, None), (2, 1)])
def test_lr_schedule_matches_torch_multisteplr(E, resume_at):
    got, want = _own_run(E, E, resume_at), _torch_run(E, E, resume_at)
    assert got == pytest.approx(want, rel=1e-12), (E, resume_at, got[-12:], want[-12:])


def test_lr_schedule_resume_needs_initial_lr():
    from speechdrivestemplates_amd.core.pipelines.voice2pose import _MultiStepLR
    with pytest.raises(KeyError):
        _MultiStepLR(_Opt(1e-4), [90, 98], 0.1, last_epoch=5)


def test_frame_variant_codes_are_rejected_like_the_reference_rejects_them():
    """VOICE2POSE.GENERATOR.CLIP_CODE.FRAME_VARIANT (voice2pose.py:66-67,148-150): the reference's own generator cannot take the (B,D,T) code
    that branch produces -- generator.py:110 calls repeat([1,1,T]) on a 4-D tensor -- so the key is dead in the reference; this build raises at
    construction.  Where the reference sources are present (the authoring container) the reference's failure is re-probed."""
    import os
    import sys
    from speechdrivestemplates_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    assert cfg.VOICE2POSE.GENERATOR.CLIP_CODE.FRAME_VARIANT is False  # the default, and no shipped yaml changes it
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speechdrivestemplates_amd", "core", "pipelines",
                            "voice2pose.py")).read()
    assert "if code.FRAME_VARIANT:" in src and "raise RuntimeError('VOICE2POSE.GENERATOR.CLIP_CODE.FRAME_VARIANT" in src
    if os.path.isdir("/root/reference/core/networks"):
        import subprocess
        code = ("import torch\nfrom types import SimpleNamespace as NS\nfrom core.networks import get_model\n"
                "cfg = NS(VOICE2POSE=NS(GENERATOR=NS(LEAKY_RELU=True, NORM='IN', CLIP_CODE=NS(DIMENSION=32))), DATASET=NS(NUM_LANDMARKS=121))\n"
                "G = get_model('SequenceGeneratorCNN')(cfg)\n"
                "try:\n    G(torch.randn(1, 80, 427), 64, torch.randn(1, 32, 64))\n    print('ACCEPTED')\n"
                "except RuntimeError as e:\n    print('RAISES', e)\n")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp",
                             env=dict(os.environ, PYTHONPATH="/root/reference", PYTHONDONTWRITEBYTECODE="1"))
        assert "RAISES" in out.stdout and "repeat dims" in out.stdout, (out.stdout, out.stderr[-500:])


def test_chain_wiring_of_the_generator():
    """ops.chain_blocks: frames / channels of the sixteen Conv1d blocks as the chain launch sees them (generator.py:53-85,96-103: e0..e6 halve the
    frames from e2 on, d5..d1 read upsample(previous) + skip, four decoder blocks) -- and inconsistent wirings raise."""
    import pytest
    from speechdrivestemplates_amd import _lib, ops
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.networks import get_model
    cfg = get_cfg_defaults()
    cfg.merge_from_list(["VOICE2POSE.GENERATOR.CLIP_CODE.DIMENSION", 32])
    net = get_model("SequenceGeneratorCNN")(cfg)
    spec, weights, slope = net._chain()
    assert len(spec) == len(weights) == 16 and slope == ops.LEAKY_SLOPE * bool(cfg.VOICE2POSE.GENERATOR.LEAKY_RELU)
    blocks = ops.chain_blocks(spec, 64, 288)
    assert [b[0] for b in blocks] == [64, 64, 64, 32, 16, 8, 4, 4, 8, 16, 32, 64, 64, 64, 64, 64]  # input frames
    assert [b[1] for b in blocks] == [64, 64, 32, 16, 8, 4, 2, 4, 8, 16, 32, 64, 64, 64, 64, 64]   # output frames
    assert [b[2] for b in blocks] == [288] + [256] * 15
    assert [b[6] for b in blocks] == [_lib.CHAIN_PLAIN] + [_lib.CHAIN_NORM] * 6 + [_lib.CHAIN_UPADD] * 5 + [_lib.CHAIN_NORM] * 4
    assert [(b[7], b[8]) for b in blocks[7:12]] == [(6, 5), (7, 4), (8, 3), (9, 2), (10, 1)]  # (upsampled block, skip block)
    for w, b in zip(weights, blocks):
        assert tuple(w.shape) == (256, b[2], b[3])
    with pytest.raises(ValueError):
        ops.chain_blocks(((4, 2, 1, _lib.CHAIN_PLAIN, -1, -1),) + tuple((4, 2, 1, _lib.CHAIN_NORM, i, -1) for i in range(7)), 64, 256)  # runs out of frames
    cfg.merge_from_list(["VOICE2POSE.GENERATOR.NORM", "BN"])
    assert get_model("SequenceGeneratorCNN")(cfg)._chain() is None  # BatchNorm generators keep the per-block kernels


def test_reducer_reserve_is_counted_not_stacked():
    """the workgroup-slot reserve of the backward stream-K plans is process-wide; reducers acquire / release it through a count, so
    dropping the FIRST of two reducers (or a late garbage collection of an old one) cannot switch it off under the one still exchanging."""
    from speechdrivestemplates_amd import dp, ops
    assert dp.active_reducers() == 0
    prev = ops.SK_RESERVED_SLOTS
    try:
        ops.SK_RESERVED_SLOTS = 0
        dp._acquire_reserve()   # reducer A
        assert ops.SK_RESERVED_SLOTS == dp.RESERVED_SLOTS
        dp._acquire_reserve()   # reducer B (created while A is alive: with a saved-value stack B would remember 32)
        dp._release_reserve()   # A goes first
        assert ops.SK_RESERVED_SLOTS == dp.RESERVED_SLOTS and dp.active_reducers() == 1
        dp._release_reserve()   # B goes: back to the single-GPU default
        assert ops.SK_RESERVED_SLOTS == 0 and dp.active_reducers() == 0
        dp._release_reserve()   # a stray extra close is harmless
        assert ops.SK_RESERVED_SLOTS == 0 and dp.active_reducers() == 0
    finally:
        ops.SK_RESERVED_SLOTS = prev


def test_three_bf16_numbers_hold_a_fp32_number_exactly():
    """The arithmetic behind the split-fp32 conv kernel (csrc/convbf.hip, x3_stage): hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) with
    round-to-nearest-even conversions and fp32 subtractions reproduce x EXACTLY (8 + 8 + 8 significand bits; |x| >= 2^-100: below that the third
    number runs out of bf16's exponent range and the split is exact to 2^-133 only), and the six products the kernel keeps
    (all but mid*lo, lo*mid, lo*lo) miss a * b by less than 2^-23 |a b|."""
    import torch
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(1 << 16, generator=g), torch.randn(1 << 12, generator=g) * 1e-30, torch.randn(1 << 12, generator=g) * 1e30,
                   torch.tensor([0.0, -0.0, 1.0, -1.0, 2.0 ** -100 * 1.2345678, 65504.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24])])

    def split(v):
        hi = v.to(torch.bfloat16).float()
        r1 = v - hi
        mid = r1.to(torch.bfloat16).float()
        r2 = r1 - mid
        lo = r2.to(torch.bfloat16).float()
        return hi, mid, lo

    hi, mid, lo = split(x)
    assert torch.equal((hi.double() + mid.double() + lo.double()).float(), x)
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double())  # exactly, not just after rounding
    a, b = x[: 1 << 16], x[: 1 << 16].flip(0)
    (ah, am, al), (bh, bm, bl) = split(a), split(b)
    kept = (al.double() * bh.double() + ah.double() * bl.double() + am.double() * bm.double() + am.double() * bh.double() + ah.double() * bm.double()
            + ah.double() * bh.double())
    exact = a.double() * b.double()
    assert ((kept - exact).abs() <= 2.0 ** -23 * exact.abs()).all()


def test_device_copies_of_speaker_statistics_are_refreshed_in_place_never_freed():
    """a captured hipGraph reads the device copy of a speaker's mean / std by raw pointer.  Registering the speaker again (every
    SyntheticGestureDataset does) or editing a registered array must keep that memory alive and put the new values INTO it."""
    import numpy as np
    from types import SimpleNamespace
    from speechdrivestemplates_amd.core.datasets import gesture_dataset as gd
    tr = gd.PoseTransforms()
    tr.cfg = SimpleNamespace(NUM_LANDMARKS=121, HIERARCHICAL_POSE=True)
    rng = np.random.default_rng(0)
    stat = {'scale_factor': 1.0, 'mean': rng.standard_normal(242), 'std': rng.random(242) + 0.5}
    gd.register_speaker_stat('_cache_test', parted=stat)
    kp = torch.zeros(2, 4, 2, 121)
    m0 = tr._stat(stat['mean'], kp)
    ptr = m0.data_ptr()
    assert tr._stat(stat['mean'], kp).data_ptr() == ptr  # served from the table
    # in-place edit: same tensor, new values
    stat['mean'][:] = stat['mean'] + 1.0
    m1 = tr._stat(stat['mean'], kp)
    assert m1.data_ptr() == ptr and torch.allclose(m1.reshape(-1), torch.tensor(stat['mean'], dtype=torch.float32))
    # the speaker is registered again with new arrays (a second dataset after a capture): the SAME memory carries the new table
    stat2 = {'scale_factor': 1.0, 'mean': rng.standard_normal(242), 'std': rng.random(242) + 0.5}
    gd.register_speaker_stat('_cache_test', parted=stat2)
    assert torch.equal(m0.reshape(-1), torch.tensor(stat2['mean'], dtype=torch.float32))  # the old view sees the new values
    assert tr._stat(stat2['mean'], kp).data_ptr() == ptr
    # filling the table past its bound never drops an entry
    keep = [rng.standard_normal(242) for _ in range(gd.PoseTransforms._STAT_CACHE_MAX + 8)]
    for a in keep:
        tr._stat(a, kp)
    assert tr._stat(stat2['mean'], kp).data_ptr() == ptr
    assert len(gd.PoseTransforms._STAT_ON_DEVICE) <= gd.PoseTransforms._STAT_CACHE_MAX
    for k in [k for k, v in gd.PoseTransforms._STAT_ON_DEVICE.items() if any(v[0] is a for a in keep)]:
        del gd.PoseTransforms._STAT_ON_DEVICE[k]  # (test hygiene only: nothing captured these)


def test_reserve_follows_the_rccl_channel_count(monkeypatch):
    """dp.reserved_slots(): 32 by default; the launcher's RCCL channel cap when there is one (a channel is one long-lived workgroup), in the plan
    builder's units (multiple of 8, at most half of the GPU); a caller's own dp.RESERVED_SLOTS wins."""
    from speechdrivestemplates_amd import dp
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("NCCL_MIN_NCHANNELS", raising=False)
    assert dp.reserved_slots() == 32
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "12")
    assert dp.reserved_slots() == 16
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "64")
    assert dp.reserved_slots() == 64
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "1000")
    assert dp.reserved_slots() == 256
    monkeypatch.setattr(dp, "RESERVED_SLOTS", 248)
    assert dp.reserved_slots() == 248
    assert isinstance(dp.other_gpu_processes(), list)
