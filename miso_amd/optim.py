"""Dense Adam as a torch.optim.Optimizer backed by miso_adam_active (miso_adam_dense's results without the
passes over elements that no gradient has ever reached: their moments are zero and their update is exactly zero).

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
    def step(self, closure=None, clear_grads=False):
        """clear_grads: leave every .grad zeroed (in the same pass for the big tensors), for callers that
        accumulate into persistent gradient buffers and would otherwise memset them before the next backward."""
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
                    if 'active' not in st:
                        # state restored from a plain Adam checkpoint: everything may already be moving
                        st['active'] = ops.adam_active_flags(p)
                        if t > 1:
                            st['active'].fill_(1)
                    ops.adam_active_(p, g, m, v, st['active'], t, lr, b1, b2, eps, zero_grad=clear_grads)
                    continue
                m.lerp_(g, 1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(m, denom, value=-(lr / bc1))
                if clear_grads:
                    g.zero_()
        return loss
