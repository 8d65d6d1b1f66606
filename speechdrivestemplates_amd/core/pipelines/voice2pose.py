"""Voice2Pose pipeline on the gfx950 engine: ``Voice2PoseModel`` (mel -> generator -> losses -> no-grad pose
encoder -> optional motion discriminator) and ``Voice2Pose.train_step`` with the reference's semantics
(core/pipelines/voice2pose.py:22-210, 216-331, 412-430).

Deliberate, documented differences from the reference:
  * the clip-code KL "skip when any batch variance is 0" test (voice2pose.py:154) stays on the device:
    ``G_clipcode_kl_loss`` is always present and is exactly 0 when skipped (``results['kl_valid']`` tells);
  * data parallelism is one summing all-reduce per optimiser group over flat gradient buffers (dp.py) instead of
    DistributedDataParallel; discriminator gradients ARE synchronised (the reference's second backward is not;
    SYS.DDP_UNSYNCED_D reproduces that);
  * per-loss scalars are reduced to rank 0 in one packed collective, only on logging steps.
"""
from collections import OrderedDict

import torch
from torch import nn

from ... import dp, ops
from ...mel import MelSpectrogram
from ...optim import FlatAdam
from ..networks import get_model
from .trainer import Trainer


class Voice2PoseModel(nn.Module):
    def __init__(self, cfg, state_dict=None, num_train_samples=None, rank=0, external_codes=None) -> None:
        super().__init__()
        self.cfg = cfg
        self.mel_transfm = MelSpectrogram(win_length=400, hop_length=160, n_fft=512, f_min=55, f_max=7500.0, n_mels=80)
        self.netG = get_model(cfg.VOICE2POSE.GENERATOR.NAME)(cfg)
        code = cfg.VOICE2POSE.GENERATOR.CLIP_CODE
        if code.DIMENSION is not None:
            if code.FRAME_VARIANT:
                # Per-frame codes (N, D, T) (voice2pose.py:66-67,148-150) cannot run in the REFERENCE either: its generator does
                # `code.unsqueeze(2).repeat([1, 1, T])` (generator.py:110), which raises for a 3-D code ("Number of dimensions of repeat
                # dims can not be smaller than number of dimensions of tensor" -- probed by importing the reference's own module,
                # tests/test_host_logic.py records it).  No shipped yaml sets the key.  Same outcome, earlier and with the reason:
                raise RuntimeError('VOICE2POSE.GENERATOR.CLIP_CODE.FRAME_VARIANT: the reference generator itself rejects per-frame codes '
                                   '(generator.py:110: repeat() on a (B,D,1,T) tensor with 3 repeat dims); unsupported here as there')
            if code.EXTERNAL_CODE:  # fixed codes from a pose-VAE checkpoint (voice2pose.py:40-55)
                if external_codes is not None:
                    self.clips_code = external_codes
                else:
                    path = code.EXTERNAL_CODE_PTH or cfg.VOICE2POSE.POSE_ENCODER.AE_CHECKPOINT
                    if path is None:
                        raise RuntimeError('External code not provide.')
                    ckpt = torch.load(path, map_location='cpu')
                    self.clips_code = {k.replace('module.', ''): v for k, v in ckpt['model_state_dict'].items()
                                       if 'clip_code' in k}['clip_code_mu']
                if num_train_samples is not None and self.clips_code.shape[0] < num_train_samples:
                    # the reference leaves this shape check commented out and fails later with an IndexError on the first
                    # clip beyond the table; fail here, before any kernel sees such an index
                    raise RuntimeError('external clip codes have %d rows but the training set has %d clips'
                                       % (self.clips_code.shape[0], num_train_samples))
            else:
                if num_train_samples is None:
                    assert state_dict is not None, 'No state_dict available, while no dataset is configured.'
                    num_train_samples = state_dict['module.clips_code'].shape[0]
                self.clips_code = nn.Parameter(torch.zeros(num_train_samples, code.DIMENSION), requires_grad=code.TRAIN)
        else:
            self.clips_code = None
        if cfg.VOICE2POSE.POSE_ENCODER.NAME is not None:
            self.pose_encoder = get_model(cfg.VOICE2POSE.POSE_ENCODER.NAME)(cfg)
            self.pose_encoder.eval()  # overridden by the trainer's model.train(), as in the reference (trainer.py:382)
        if cfg.VOICE2POSE.POSE_DISCRIMINATOR.NAME is not None:
            self.netD_pose = get_model(cfg.VOICE2POSE.POSE_DISCRIMINATOR.NAME)(cfg)

    # ---------------------------------------------------------------------------------------------
    def _device(self):
        return self.netG.decoder[4].weight.device

    def _code_table(self, dev):
        if self.clips_code.device != dev:  # plain-tensor external codes are not moved by .cuda()
            self.clips_code = self.clips_code.to(dev)
        return self.clips_code

    def _eval_code(self, batch, n, dev, poses_gt, dataset, speaker, interpolation_coeff, return_loss):
        """Code selection outside training (voice2pose.py:95-120)."""
        code = self.cfg.VOICE2POSE.GENERATOR.CLIP_CODE
        if code.SAMPLE_FROM_NORMAL:
            return torch.randn([n, code.DIMENSION], device=dev)
        if code.TEST_WITH_GT_CODE:
            assert self.cfg.VOICE2POSE.POSE_ENCODER.NAME is not None
            src = poses_gt if self.cfg.DATASET.HIERARCHICAL_POSE else dataset.transform_normalized_parted2global(poses_gt, speaker)
            with torch.no_grad():
                return self.pose_encoder(src)[0]
        table = self._code_table(dev)
        if self.cfg.DEMO.CODE_INDEX is not None:
            assert not return_loss, 'WARNING: Do not set "DEMO.CODE_INDEX" in train or test mode!'
            assert 0 <= self.cfg.DEMO.CODE_INDEX < table.size(0)
            c = table[torch.full((n,), self.cfg.DEMO.CODE_INDEX, dtype=torch.long, device=dev)]
            if interpolation_coeff is not None:
                assert self.cfg.DEMO.CODE_INDEX_B < table.size(0)
                cb = table[torch.full((n,), self.cfg.DEMO.CODE_INDEX_B, dtype=torch.long, device=dev)]
                c = c * (1 - interpolation_coeff) + cb * interpolation_coeff
            return c
        return table[torch.randint(table.size(0), (n,), device=dev)]

    def forward(self, batch, dataset, return_loss=True, interpolation_coeff=None):
        cfg = self.cfg
        g = cfg.VOICE2POSE.GENERATOR
        dev = self._device()
        audio = batch['audio'].to(dev, non_blocking=True)
        speaker = batch['speaker']
        clip_indices = batch['clip_index'].to(dev, non_blocking=True)
        num_frames = int(batch['num_frames'][0])
        poses_gt = batch['poses'].to(dev, non_blocking=True) if return_loss else None

        kl = kl_valid = None
        if g.CLIP_CODE.DIMENSION is not None:
            if self.training:
                condition_code, kl, kl_valid = ops.CodeGatherKLFn.apply(self._code_table(dev), clip_indices, g.LAMBDA_CLIP_KL)
            else:
                condition_code = self._eval_code(batch, audio.shape[0], dev, poses_gt, dataset, speaker, interpolation_coeff,
                                                 return_loss)
                if return_loss:
                    ar = torch.arange(condition_code.shape[0], device=dev)
                    _, kl, kl_valid = ops.CodeGatherKLFn.apply(condition_code.detach().contiguous(), ar, g.LAMBDA_CLIP_KL, True)
        else:
            condition_code = None

        mel = self.mel_transfm(audio)
        poses_pred = self.netG(mel, num_frames, condition_code)
        results = {'poses_pred_batch': poses_pred, 'condition_code': condition_code}
        if not return_loss:
            return results
        results['poses_gt_batch'] = poses_gt

        losses = {}
        reg = ops.L1LossFn.apply(poses_pred, poses_gt, float(g.LAMBDA_REG))  # voice2pose.py:141-142
        losses['G_reg_loss'] = reg
        g_loss = reg
        if kl is not None:
            losses['G_clipcode_kl_loss'] = kl
            results['kl_valid'] = kl_valid
            g_loss = g_loss + kl
        losses['G_loss'] = g_loss

        if cfg.VOICE2POSE.POSE_ENCODER.NAME is not None:  # FGD features, off the loss path (voice2pose.py:160-176)
            # These 14 small conv+BN launches are latency-bound and feed only results_dict: with ops.OVERLAP_AUX they
            # run on a side stream, concurrently with the MFMA-bound backward pass that follows on the main stream
            # (ops.join_side_stream() before the optimiser step / before the results are read).
            side = ops.side_stream_scope(self.training and torch.is_grad_enabled())
            with side, torch.no_grad():
                if cfg.DATASET.HIERARCHICAL_POSE:
                    e_pred, e_gt = poses_pred.detach(), poses_gt
                else:
                    e_pred = dataset.transform_normalized_parted2global(poses_pred.detach().clone(), speaker)
                    e_gt = dataset.transform_normalized_parted2global(poses_gt.clone(), speaker)
                side.uses(e_pred, e_gt)
                ops.stage_mark("pose_enc:begin")
                if ops.PAIR_AUX:  # one pass over the concatenated batch, statistics and running-statistics updates per half (= the two calls)
                    (mu_pred, logvar_pred), (mu_gt, logvar_gt) = self.pose_encoder.forward_pair(e_pred, e_gt)
                else:
                    mu_pred, logvar_pred = self.pose_encoder(e_pred)
                    mu_gt, logvar_gt = self.pose_encoder(e_gt)
                ops.stage_mark("pose_enc:end")
            results.update(mu_pred=mu_pred, mu_gt=mu_gt, logvar_pred=logvar_pred, logvar_gt=logvar_gt)

        if hasattr(self, 'netD_pose'):  # LSGAN on motion patches (voice2pose.py:179-208)
            d = cfg.VOICE2POSE.POSE_DISCRIMINATOR
            B, T = poses_gt.shape[0], poses_gt.shape[1]
            real, fake = poses_gt, poses_pred
            if d.WHITE_LIST is not None:
                real, fake = real[..., d.WHITE_LIST], fake[..., d.WHITE_LIST]
            if d.MOTION:
                real = ops.TimeDiffFn.apply(real.reshape(B, T, -1)).reshape(B, T - 1, 2, -1)
                fake = ops.TimeDiffFn.apply(fake.reshape(B, T, -1)).reshape(B, T - 1, 2, -1)
            s_real = self.netD_pose(real)
            s_fake = self.netD_pose(fake)
            s_fake_det = self.netD_pose(fake.detach())
            g_gan = ops.MseConstFn.apply(s_fake, 1.0, d.LAMBDA_GAN)  # nn.MSELoss against ones / zeros (voice2pose.py:189-197)
            losses['G_pose_gan_loss'] = g_gan
            losses['G_loss'] = g_loss + g_gan
            d_loss = ops.MseConstFn.apply(s_real, 1.0, d.LAMBDA_GAN) + ops.MseConstFn.apply(s_fake_det, 0.0, d.LAMBDA_GAN)
            losses.update(D_pose_gan_loss=d_loss, pose_score_fake=s_fake.mean(), pose_score_real=s_real.mean())
        return losses, results


class Voice2Pose(Trainer):
    def __init__(self, cfg) -> None:
        super().__init__(cfg)

    def setup_model(self, cfg, state_dict=None, external_codes=None):
        # kernel routing this pipeline's configuration asks for -- set UNCONDITIONALLY (defaults included: a pipeline built after a bf16 one in the
        # same process must not inherit its mode) and re-applied at the start of every step this pipeline runs (Trainer.apply_knobs).
        # Before setup_optimizer: its weight mirrors allocate the bf16 copies.
        self.knobs = {'storage': getattr(cfg.SYS, 'STORAGE', 'f32'), 'chain1d': bool(getattr(cfg.SYS, 'CHAIN1D', True)),
                      'f32_split': bool(getattr(cfg.SYS, 'CONV_F32_SPLIT', True))}
        self.apply_knobs()
        self.model = Voice2PoseModel(cfg, state_dict, self.num_train_samples, self.get_rank(), external_codes).cuda()
        if state_dict is not None:
            sd = OrderedDict((k[len('module.'):] if k.startswith('module.') else k, v) for k, v in state_dict.items())
            self.model.load_state_dict(sd, strict=bool(cfg.VOICE2POSE.STRICT_LOADING))
        if cfg.VOICE2POSE.POSE_ENCODER.NAME is not None and cfg.VOICE2POSE.POSE_ENCODER.AE_CHECKPOINT is not None:
            ckpt = torch.load(cfg.VOICE2POSE.POSE_ENCODER.AE_CHECKPOINT, map_location='cpu')  # voice2pose.py:235-242
            enc = OrderedDict((k.replace('module.ae.encoder.', ''), v) for k, v in ckpt['model_state_dict'].items() if 'encoder' in k)
            self.model.pose_encoder.load_state_dict(enc)

    def setup_optimizer(self, checkpoint=None, last_epoch=-1):
        """Adam for netG, netD_pose and the clip-code table (voice2pose.py:244-279), each over one flat buffer."""
        cfg = self.cfg
        E = cfg.TRAIN.NUM_EPOCHS

        def add(name, params, lr, wd=0):
            opt = FlatAdam(params, lr=lr, weight_decay=wd)
            if checkpoint is not None:
                opt.load_state_dict(checkpoint[name + '_state_dict'])
            self.optimizers[name] = opt
            if cfg.TRAIN.LR_SCHEDULER:
                self.schedulers[name.replace('optimizer', 'scheduler')] = _MultiStepLR(opt, [E - 10, E - 2], 0.1, last_epoch)

        add('optimizerG', self.model.netG.parameters(), cfg.TRAIN.LR, cfg.TRAIN.WD)
        if cfg.VOICE2POSE.POSE_DISCRIMINATOR.NAME is not None:
            add('optimizerD_pose', self.model.netD_pose.parameters(), cfg.TRAIN.LR)
        code = cfg.VOICE2POSE.GENERATOR.CLIP_CODE
        if code.DIMENSION is not None and not code.EXTERNAL_CODE and code.TRAIN:
            add('optimizerClipCode', [self.model.clips_code], cfg.TRAIN.LR * code.LR_SCALING)
        self._set_reducer(dp.GradReducer(self.optimizers.values()))
        # the reference's quirk D8 behind a flag: the discriminator's second backward is not exchanged -- local gradients, no 1 / world_size
        self.unsynced_d = bool(getattr(cfg.SYS, 'DDP_UNSYNCED_D', False)) and 'optimizerD_pose' in self.optimizers
        if self.unsynced_d:
            self.optimizers['optimizerD_pose'].grad_scale = 1.0
        # DDP-constructor semantics (voice2pose.py:222-223): every rank starts from rank 0's parameters, buffers and Adam state
        dp.sync_replicas(self.model, list(self.optimizers.values()))
        if self.reducer.active:
            # Gradient buckets in the order backward completes them (the flat buffer is laid out audio encoder L0..L7,
            # U-Net, decoder): [U-Net + decoder, ~14 MB] when backward reaches the audio encoder, then [L5..L7, ~11 MB],
            # [L3..L4, 2.2 MB] and [L1..L2, 0.5 MB] as backward leaves those blocks; only L0 (2.3 KB) and the code table
            # (0.5 MB) go out after backward, right before the optimiser step.
            optg = self.optimizers['optimizerG']
            names = [n for n, p in self.model.netG.named_parameters() if p.requires_grad]
            first = next((i for i, n in enumerate(names) if not n.startswith('audio_encoder.')), None)
            if first is not None and all(not n.startswith('audio_encoder.') for n in names[first:]):
                lo, reducer = optg.offsets[first], self.reducer

                def _launch_late_layers(grad, _optg=optg, _lo=lo, _r=reducer):
                    _r.launch(_optg, _lo, None)
                    return None

                self.model.netG.post_encoder_grad_hook = _launch_late_layers
                hooks, hi = {}, lo
                for blk in self.ENCODER_BUCKET_BLOCKS:  # descending: backward order
                    pre = 'audio_encoder.specgram_encoder_2d.%d.%d.' % (blk // 2, blk % 2)
                    i0 = next((i for i, n in enumerate(names) if n.startswith(pre)), None)
                    if i0 is None or not all(n.startswith('audio_encoder.') for n in names[i0:first]):
                        break
                    lo_b = optg.offsets[i0]

                    def _launch_block_range(grad, _optg=optg, _lo=lo_b, _hi=hi, _r=reducer):
                        _r.launch(_optg, _lo, _hi)
                        return None

                    hooks[blk] = _launch_block_range
                    hi = lo_b
                self.model.netG.audio_encoder.grad_bucket_hooks = hooks

    ENCODER_BUCKET_BLOCKS = (5, 3, 1)  # a hook on the INPUT gradient of these audio-encoder blocks closes a bucket

    # ---------------------------------------------------------------------------------------------
    def forward_backward(self, batch, want_final=False):
        """Forward, per-step metrics and both backward passes (voice2pose.py:288-301,306-308) -- everything of a
        train step up to (not including) the gradient exchange and the optimiser updates."""
        dev = self.model._device()
        self.apply_knobs()
        ops.begin_step(dev)
        losses, results = self.model(batch, self.train_dataset)
        stat = batch['speaker_stat']
        pred, gt = results['poses_pred_batch'].detach(), results['poses_gt_batch']
        mean, std = stat['mean'].to(dev, non_blocking=True), stat['std'].to(dev, non_blocking=True)
        scale = stat['scale_factor'].to(dev, non_blocking=True)
        # the float64 per-step metrics feed only the log: side stream, concurrent with backward (joined before the
        # optimiser step, like the pose-encoder passes)
        side = ops.side_stream_scope(True)
        with side:
            side.uses(pred, gt, mean, std, scale)
            fin_p, fin_g, metrics = ops.final_metrics(pred, gt, mean, std, scale, bool(self.cfg.DATASET.HIERARCHICAL_POSE),
                                                      want_final)
        results['poses_pred_normalized'] = results['poses_pred_batch']  # extension: the raw network output
        if want_final:
            results['poses_pred_batch'], results['poses_gt_batch'] = fin_p, fin_g
        losses['L2_dist'], losses['lip_sync_error_n'] = metrics[0], metrics[1]
        has_d = 'optimizerD_pose' in self.optimizers
        if 'optimizerClipCode' in self.optimizers:
            self.optimizers['optimizerClipCode'].zero_grad()
        self.optimizers['optimizerG'].zero_grad()
        ops.defer_small_dw(True)  # the 1-D stage's weight gradients go to the side stream in one batch (ops.flush_deferred_dw)
        try:
            losses['G_loss'].backward(retain_graph=has_d)
        finally:
            ops.defer_small_dw(False)
            ops.flush_deferred_dw()
        return losses, results

    def optimizer_updates(self, losses):
        """Gradient all-reduce + Adam steps (voice2pose.py:302-309)."""
        has_d = 'optimizerD_pose' in self.optimizers
        group = [self.optimizers[k] for k in ('optimizerClipCode', 'optimizerG') if k in self.optimizers]
        self.reducer.all_reduce(group)
        for opt in group:
            opt.step()
        if has_d:
            optd = self.optimizers['optimizerD_pose']
            optd.zero_grad()
            losses['D_pose_gan_loss'].backward()
            if not getattr(self, 'unsynced_d', False):  # (SYS.DDP_UNSYNCED_D: the reference's second backward is not exchanged)
                self.reducer.all_reduce([optd])
            optd.step()

    def train_step(self, batch, t_step, global_step, epoch):
        tag = 'TRAIN'
        log_step = t_step % self.cfg.SYS.LOG_INTERVAL == 0
        save_step = t_step % self.result_saving_interval_train == 0 and (self.cfg.TRAIN.SAVE_NPZ or self.cfg.TRAIN.SAVE_VIDEO)
        # SYS.HIP_GRAPH: replay the captured step (forward, metrics, backward, gradient exchange, Adam): the host copies the batch into the graph's
        # static inputs and launches ONE graph instead of ~270 kernels -- on one GPU and under data parallelism alike (graph.GraphedStep captures the
        # RCCL all-reduces with the step, or replays two graphs around an eager exchange; steps that save results run eagerly: they need the final poses)
        losses, results = self.graphed_or_eager_step(batch, eager_ok=not save_step, want_final=bool(save_step))
        self.last_losses = losses
        if log_step:
            if self.cfg.SYS.DISTRIBUTED:
                # the per-rank kernel error words travel with the loss scalars: every rank stops HERE when any rank's launch lost a partner
                self.check_kernels_all_ranks(dp.reduce_scalars(losses, error_flag=ops.kernel_error_flag()))
            if self.is_master_process():
                self.logger_writer_step(tag, losses, t_step, epoch, global_step)
        if save_step and self.is_master_process() and self.cfg.TRAIN.SAVE_NPZ:
            self.save_results(tag, t_step, epoch, self.base_path,
                              {k: v.detach().cpu().numpy() for k, v in results.items() if torch.is_tensor(v)})

    @torch.no_grad()
    def test_step(self, batch, t_step, epoch=0):
        """Validation / test step (voice2pose.py:333-384) without the video writer."""
        tag = 'TEST' if epoch == 0 else 'VAL'
        dev = self.model._device()
        self.apply_knobs()
        m = self.cfg.TEST.MULTIPLE
        assert isinstance(m, int) and m >= 1, 'TEST.MULTIPLE should be an integer that larger than 1, but get %r (%s).' % (m, type(m))
        if m > 1:
            batch = self.mutiply_batch(batch, m)
        losses, results = self.model(batch, self.test_dataset)
        stat = batch['speaker_stat']
        fin_p, fin_g, metrics = ops.final_metrics(results['poses_pred_batch'], results['poses_gt_batch'], stat['mean'].to(dev),
                                                  stat['std'].to(dev), stat['scale_factor'].to(dev),
                                                  bool(self.cfg.DATASET.HIERARCHICAL_POSE), True)
        results['poses_pred_batch'], results['poses_gt_batch'] = fin_p, fin_g
        losses['L2_dist'], losses['lip_sync_error_n'] = metrics[0], metrics[1]
        if self.cfg.SYS.DISTRIBUTED:
            dp.reduce_scalars(losses)
        if self.is_master_process():
            if t_step % self.cfg.SYS.LOG_INTERVAL == 0:
                self.logger_writer_step(tag, losses, t_step, epoch)
            if t_step % self.result_saving_interval_test == 0 and self.cfg.TEST.SAVE_NPZ and self.base_path is not None:
                self.save_results(tag, t_step, epoch, self.base_path,
                                  {k: v.detach().cpu().numpy() for k, v in results.items() if torch.is_tensor(v)})
        batch_losses = {k: v.detach() * self.cfg.TEST.BATCH_SIZE for k, v in losses.items()}
        keep = ('mu_pred', 'mu_gt', 'logvar_pred', 'logvar_gt', 'condition_code')
        return batch_losses, {k: v.detach().cpu().numpy() for k, v in results.items() if k in keep and v is not None}

    @torch.no_grad()
    def demo_step(self, batch, t_step=0, epoch=0, extra_id=None, interpolation_coeff=None):
        """Variable-length inference from raw audio (voice2pose.py:386-410 without the video writer): batch['audio'] is
        (1, L) with L cropped to a whole number of 1/15 s frames, batch['num_frames'] = L // (16000/15) (up to 360 for the
        reference's 24 s demo limit); returns de-normalised global poses (1, T, 2, 121) in float64."""
        self.model.eval()
        self.apply_knobs()
        results = self.model(batch, self.test_dataset, return_loss=False, interpolation_coeff=interpolation_coeff)
        results['poses_pred_batch'] = self.test_dataset.get_final_results(results['poses_pred_batch'].detach(), batch['speaker_stat'])
        if self.is_master_process() and self.cfg.TEST.SAVE_NPZ and self.base_path is not None:
            self.save_results('DEMO', t_step, epoch, self.base_path,
                              {k: v.detach().cpu().numpy() for k, v in results.items() if torch.is_tensor(v)}, extra_id=extra_id)
        return results

    def evaluate_step(self, results_dict):
        """L2 distance and normalised lip-sync error (voice2pose.py:412-430) on final (de-normalised) poses."""
        p, g = results_dict['poses_pred_batch'], results_dict['poses_gt_batch']
        l2 = torch.norm(p - g, p=2, dim=2)
        lp = torch.norm(p[:, :, :, 75] - p[:, :, :, 71], p=2, dim=-1)
        lg = torch.norm(g[:, :, :, 75] - g[:, :, :, 71], p=2, dim=-1)
        den = lg.max(-1, keepdim=True).values + 1e-4
        return {'L2_dist': l2.mean(), 'lip_sync_error_n': torch.abs(lp / den - lg / den).mean()}

    def evaluate_epoch(self, results_dict):
        from ...fgd import compute_fgd
        import numpy as np
        return {'FGD_mu': compute_fgd(results_dict['mu_pred'], results_dict['mu_gt']),
                'FGD_mu_logvar': compute_fgd(np.concatenate([results_dict['mu_pred'], results_dict['logvar_pred']], axis=1),
                                             np.concatenate([results_dict['mu_gt'], results_dict['logvar_gt']], axis=1))}


class _MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR on FlatAdam.param_groups, stepped once per epoch (voice2pose.py:253-279,
    trainer.py:396-398), with torch's exact (recursive) semantics, which the reference's resume path depends on:
      * construction does one ``step()``; a step multiplies the CURRENT lr by gamma^(multiplicity) only when the new
        ``last_epoch`` EQUALS a milestone -- a milestone <= 0 other than 0 never fires, and a resumed optimiser keeps the
        (already decayed) lr it was saved with;
      * resuming (``last_epoch != -1``) requires 'initial_lr' in the loaded param group (torch raises KeyError otherwise);
      * resumed from a checkpoint of epoch k the counter sits at k+1 while epoch k runs, so milestones fire one epoch
        earlier than in an uninterrupted run (reference behaviour, reproduced)."""

    def __init__(self, opt, milestones, gamma, last_epoch=-1):
        self.opt, self.gamma = opt, gamma
        self.milestones = {m: list(milestones).count(m) for m in milestones}
        g = opt.param_groups[0]
        if last_epoch == -1:
            g.setdefault('initial_lr', g['lr'])
        elif 'initial_lr' not in g:
            raise KeyError("param 'initial_lr' is not specified in param_groups[0] when resuming an optimizer")
        self.base_lr = g['initial_lr']
        self.last_epoch = last_epoch
        self.step()

    def step(self):
        self.last_epoch += 1
        if self.last_epoch in self.milestones:
            self.opt.param_groups[0]['lr'] *= self.gamma ** self.milestones[self.last_epoch]
        self.opt.sync_lr()
