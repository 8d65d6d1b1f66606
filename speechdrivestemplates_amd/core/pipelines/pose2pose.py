"""Pose2Pose (pose-sequence VAE) pipeline on the gfx950 engine -- reference core/pipelines/pose2pose.py:20-169:
L1 reconstruction + analytic KL on (mu, logvar), one Adam, per-clip code buffers written every step."""
import numpy as np
import torch
from torch import nn

from ... import dp, ops
from ...mel import MelSpectrogram
from ...optim import FlatAdam
from ..networks import get_model
from .trainer import Trainer
from .voice2pose import _MultiStepLR


class Pose2PoseModel(nn.Module):
    def __init__(self, cfg, state_dict=None, num_train_samples=None, rank=0) -> None:
        super().__init__()
        self.cfg = cfg
        self.mel_transfm = MelSpectrogram(win_length=400, hop_length=160, n_fft=512, f_min=55, f_max=7500.0, n_mels=80)
        self.ae = get_model(cfg.POSE2POSE.AUTOENCODER.NAME)(cfg)
        if num_train_samples is None:
            assert state_dict is not None, 'No state_dict available, while no dataset is configured.'
            num_train_samples = state_dict['module.clip_code_mu'].shape[0]
        d = cfg.POSE2POSE.AUTOENCODER.CODE_DIM
        self.register_buffer('clip_code_mu', torch.zeros([num_train_samples, d]))
        self.register_buffer('clip_code_logvar', torch.zeros([num_train_samples, d]))

    def forward(self, batch, return_loss=True, is_testing=False, interpolation_coeff=None):
        dev = self.clip_code_mu.device
        poses_gt = batch['poses'].to(dev, non_blocking=True) if return_loss else None
        num_frames = int(batch['num_frames'][0])
        if not return_loss:  # decode one stored code, picked by the interpolation coefficient (pose2pose.py:50-63)
            assert self.cfg.DEMO.CODE_PATH is not None
            idx = int((self.cfg.DEMO.MULTIPLE - 1) * interpolation_coeff)
            code = np.load(self.cfg.DEMO.CODE_PATH)['v'][idx] * 10  # "10 is empirically selected" (the reference's words)
            code = torch.tensor(np.asarray(code), dtype=torch.float32, device=dev).unsqueeze(0)
            pred, mu, logvar = self.ae(None, self.cfg.DATASET.NUM_FRAMES, external_code=code)
            return {'poses_pred_batch': pred, 'clip_code_mu': mu, 'clip_code_logvar': logvar}
        # the reference computes the mel spectrogram here and discards it (pose2pose.py:48, autoencoder.py:79); skipped
        pred, mu, logvar = self.ae(poses_gt, num_frames)
        reg = ops.L1LossFn.apply(pred, poses_gt, float(self.cfg.POSE2POSE.LAMBDA_REG))
        kl = 0.5 * (-logvar + mu ** 2 + torch.exp(logvar) - 1).mean() * self.cfg.POSE2POSE.LAMBDA_KL  # pose2pose.py:76
        losses = {'reg_loss': reg, 'kl_loss': kl, 'loss': reg + kl}
        return losses, {'poses_pred_batch': pred, 'poses_gt_batch': poses_gt, 'clip_code_mu': mu, 'clip_code_logvar': logvar}


class Pose2Pose(Trainer):
    def __init__(self, cfg) -> None:
        super().__init__(cfg)

    def setup_model(self, cfg, state_dict=None):
        # the pose VAE is 1-D only: fp32 tensors (Trainer.apply_knobs); every knob explicit, so that nothing is inherited from a pipeline that ran before
        self.knobs = {'storage': 'f32', 'chain1d': bool(getattr(cfg.SYS, 'CHAIN1D', True)), 'f32_split': bool(getattr(cfg.SYS, 'CONV_F32_SPLIT', True))}
        self.apply_knobs()
        self.model = Pose2PoseModel(cfg, state_dict, self.num_train_samples, self.get_rank()).cuda()
        if state_dict is not None:
            self.model.load_state_dict({(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()})

    def setup_optimizer(self, checkpoint=None, last_epoch=-1):
        opt = FlatAdam(self.model.ae.parameters(), lr=self.cfg.TRAIN.LR, weight_decay=self.cfg.TRAIN.WD)
        if checkpoint is not None:
            opt.load_state_dict(checkpoint['optimizer_state_dict'])
        self.optimizers['optimizer'] = opt
        if self.cfg.TRAIN.LR_SCHEDULER:
            E = self.cfg.TRAIN.NUM_EPOCHS
            self.schedulers['scheduler'] = _MultiStepLR(opt, [E - 10, E - 2], 0.1, last_epoch)
        self._set_reducer(dp.GradReducer(self.optimizers.values()))
        dp.sync_replicas(self.model, list(self.optimizers.values()))  # DDP-constructor semantics (pose2pose.py:102)

    def forward_backward(self, batch, want_final=False):
        dev = self.model.clip_code_mu.device
        self.apply_knobs()
        ops.begin_step(dev)
        losses, results = self.model(batch)
        stat = batch['speaker_stat']
        _, _, metrics = ops.final_metrics(results['poses_pred_batch'].detach(), results['poses_gt_batch'], stat['mean'].to(dev),
                                          stat['std'].to(dev), stat['scale_factor'].to(dev),
                                          bool(self.cfg.DATASET.HIERARCHICAL_POSE), False)
        idx = batch['clip_index'].to(dev)
        self.model.clip_code_mu[idx] = results['clip_code_mu'].detach()  # pose2pose.py:135-137
        self.model.clip_code_logvar[idx] = results['clip_code_logvar'].detach()
        losses['L2_dist'], losses['lip_sync_error_n'] = metrics[0], metrics[1]
        opt = self.optimizers['optimizer']
        opt.zero_grad()
        ops.defer_small_dw(True)  # every weight gradient of this model is small: one grouped launch at the end of backward (ops.flush_deferred_dw)
        try:
            losses['loss'].backward()
        finally:
            ops.defer_small_dw(False)
            ops.flush_deferred_dw()
        return losses, results

    def optimizer_updates(self, losses):
        opt = self.optimizers['optimizer']
        self.reducer.all_reduce([opt])
        opt.step()

    def train_step(self, batch, t_step, global_step, epoch):
        # SYS.HIP_GRAPH: ~150 launches of a few microseconds -- enqueued one by one the step is bound by the host (3 ms); replayed from a hipGraph
        # (one GPU or data-parallel: graph.GraphedStep) it is not
        losses, _ = self.graphed_or_eager_step(batch)
        self.last_losses = losses
        if t_step % self.cfg.SYS.LOG_INTERVAL == 0:
            if self.cfg.SYS.DISTRIBUTED:
                self.check_kernels_all_ranks(dp.reduce_scalars(losses, error_flag=ops.kernel_error_flag()))
            if self.is_master_process():
                self.logger_writer_step('TRAIN', losses, t_step, epoch, global_step)

    @torch.no_grad()
    def test_step(self, batch, t_step, epoch=0):
        """Validation / test step of the pose VAE (pose2pose.py:172-217) without the video writer."""
        tag = 'TEST' if epoch == 0 else 'VAL'
        dev = self.model.clip_code_mu.device
        self.apply_knobs()
        m = self.cfg.TEST.MULTIPLE
        assert isinstance(m, int) and m >= 1, 'TEST.MULTIPLE should be an integer that larger than 1, but get %r (%s).' % (m, type(m))
        if m > 1:
            batch = self.mutiply_batch(batch, m)
        losses, results = self.model(batch, is_testing=True)
        stat = batch['speaker_stat']
        fin_p, fin_g, metrics = ops.final_metrics(results['poses_pred_batch'], results['poses_gt_batch'], stat['mean'].to(dev),
                                                  stat['std'].to(dev), stat['scale_factor'].to(dev),
                                                  bool(self.cfg.DATASET.HIERARCHICAL_POSE), True)
        results['poses_pred_batch'], results['poses_gt_batch'] = fin_p, fin_g
        losses['L2_dist'], losses['lip_sync_error_n'] = metrics[0], metrics[1]
        if m > 1:  # spread of the per-copy mean distance over the TEST.MULTIPLE stochastic decodings (pose2pose.py:271-281)
            per_copy = torch.norm(fin_p - fin_g, p=2, dim=2).reshape(m, -1).mean(1)
            losses['L2_dist_min'], losses['L2_dist_max'] = per_copy.min(), per_copy.max()
        if self.cfg.SYS.DISTRIBUTED:
            dp.reduce_scalars(losses)
        if self.is_master_process():
            if t_step % self.cfg.SYS.LOG_INTERVAL == 0:
                self.logger_writer_step(tag, losses, t_step, epoch)
            if t_step % self.result_saving_interval_test == 0 and self.cfg.TEST.SAVE_NPZ and self.base_path is not None:
                self.save_results(tag, t_step, epoch, self.base_path,
                                  {k: v.detach().cpu().numpy() for k, v in results.items() if torch.is_tensor(v)})
        return {k: v.detach() * self.cfg.TEST.BATCH_SIZE for k, v in losses.items()}, {}


    @torch.no_grad()
    def demo_step(self, batch, t_step=0, epoch=0, extra_id=None, interpolation_coeff=None):
        """Decode a stored clip code into a pose sequence (pose2pose.py:219-244) without the video writer."""
        self.model.eval()
        results = self.model(batch, return_loss=False, interpolation_coeff=interpolation_coeff)
        results['poses_pred_batch'] = self.test_dataset.get_final_results(results['poses_pred_batch'].detach(), batch['speaker_stat'])
        if self.is_master_process() and self.cfg.TEST.SAVE_NPZ and self.base_path is not None:
            self.save_results('DEMO', t_step, epoch, self.base_path,
                              {k: v.detach().cpu().numpy() for k, v in results.items() if torch.is_tensor(v)}, extra_id=extra_id)
        return results
