"""Flat-buffer Adam: every parameter of an optimiser group is re-homed into ONE contiguous fp32 buffer (and its
gradient into a second one), so that a step is a single fused kernel launch and a data-parallel gradient
exchange is a single all-reduce.  Same update rule / defaults as ``torch.optim.Adam`` as the reference uses it
(core/pipelines/voice2pose.py:249-279): dense updates, so rows of the clip-code table with zero gradient still
have their moments decayed.  Learning rate and step counter live on the device (hipGraph-safe)."""
import torch

from . import ops


def _physical_perm(t):
    return sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))


class FlatAdam:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on the GPU only')
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned inside the flat buffers
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros_like(self.flat_param)
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self._perms = []
        for p, off in zip(self.params, self.offsets):
            perm = _physical_perm(p.data)
            phys = p.data.permute(perm)
            if not phys.is_contiguous():
                raise RuntimeError('parameter is not dense in memory')
            inv = [perm.index(d) for d in range(p.dim())]
            view = self.flat_param[off:off + p.numel()].view(phys.shape)
            view.copy_(phys)
            p.data = view.permute(inv)
            gview = self.flat_grad[off:off + p.numel()].view(phys.shape).permute(inv)
            if p.grad is not None:
                gview.copy_(p.grad)
            p.grad = gview
            self._perms.append((perm, inv, tuple(phys.shape)))
        self.param_groups = [dict(params=self.params, lr=float(lr), betas=tuple(betas), eps=float(eps),
                                  weight_decay=float(weight_decay))]
        self.lr_dev = torch.tensor([float(lr)], device=dev, dtype=torch.float32)
        self._lr_host = float(lr)
        self.state_dev = torch.zeros(2, device=dev, dtype=torch.int64)  # {int64 step; float bc1; float bc2_sqrt}
        self.grad_scale = 1.0
        self.mirrors = ops.WeightMirrors(self.params)  # transposed conv weights for the input-gradient kernels

    # -- torch.optim.Optimizer surface used by the reference -------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()

    def sync_lr(self):
        lr = float(self.param_groups[0]['lr'])
        if lr != self._lr_host:  # MultiStepLR edits param_groups between epochs (trainer.py:396-398)
            self.lr_dev.fill_(lr)
            self._lr_host = lr

    def step(self):
        g = self.param_groups[0]
        ops.join_side_stream()  # weight-gradient kernels may still be running on the side stream (ops.OVERLAP_DW)
        self.sync_lr()
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr_dev, self.state_dev,
                      g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.grad_scale)
        self.mirrors.mark_dirty()  # the next backward pass refreshes all mirrors in one launch

    def _per_param(self, flat, i):
        p, off = self.params[i], self.offsets[i]
        perm, inv, shape = self._perms[i]
        return flat[off:off + p.numel()].view(shape).permute(inv)

    def state_dict(self):
        """torch.optim.Adam-compatible layout (the reference checkpoints '<name>_state_dict', trainer.py:318-319)."""
        step = int(self.state_dev[0].item())
        state = {}
        if step > 0:
            for i in range(len(self.params)):
                state[i] = {'step': torch.tensor(float(step)), 'exp_avg': self._per_param(self.exp_avg, i).clone(),
                            'exp_avg_sq': self._per_param(self.exp_avg_sq, i).clone()}
        g = dict(self.param_groups[0])
        g['params'] = list(range(len(self.params)))
        g.update(amsgrad=False, maximize=False)
        return {'state': state, 'param_groups': [g]}

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        for k in ('lr', 'betas', 'eps', 'weight_decay', 'initial_lr'):  # 'initial_lr' is what an lr scheduler left there
            if k in g:
                self.param_groups[0][k] = g[k]
        step = 0
        for i, st in sd['state'].items():
            i = int(i)
            self._per_param(self.exp_avg, i).copy_(st['exp_avg'])
            self._per_param(self.exp_avg_sq, i).copy_(st['exp_avg_sq'])
            step = int(st['step']) if not torch.is_tensor(st['step']) else int(st['step'].item())
        self.state_dev.zero_()
        self.state_dev[0] = step
        self.sync_lr()
