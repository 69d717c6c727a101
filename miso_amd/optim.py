"""Dense Adam as a torch.optim.Optimizer backed by miso_adam_dense.

Same update and state layout ('step', 'exp_avg', 'exp_avg_sq') as
torch.optim.Adam(amsgrad=False, weight_decay=0), which is what the reference's
trainers and alignment loops construct (grid_opt/trainer.py:96-97, :424-437;
grid_opt/align/base.py:110-111).  Big dense tensors (the feature grids) take one
HIP launch each; tiny ones (pose vectors) and CPU tensors use the identical
formula in a few torch ops.  Parameters whose .grad is None are skipped, as torch does."""
import math

import torch

from . import ops

_KERNEL_MIN_NUMEL = 4096


class DenseAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (b1, b2), eps = group['lr'], group['betas'], group['eps']
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                t, m, v = st['step'], st['exp_avg'], st['exp_avg_sq']
                if (p.is_cuda and p.dtype == torch.float32 and p.numel() >= _KERNEL_MIN_NUMEL
                        and g.stride() == p.stride() and m.stride() == p.stride()):
                    ops.adam_dense_(p, g, m, v, t, lr, b1, b2, eps)
                    continue
                m.lerp_(g, 1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(m, denom, value=-(lr / bc1))
        return loss
