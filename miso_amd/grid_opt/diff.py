"""Spatial gradient of a scalar field (reference: grid_opt/diff.py:14-38)."""
import torch


def gradient3d(x, f, method='finitediff', finite_diff_eps=1e-2, create_graph=True):
    assert x.ndim == 2 and x.shape[-1] == 3
    if method == 'finitediff':
        cols = []
        for axis in range(3):
            e = torch.zeros(3, device=x.device, dtype=x.dtype)
            e[axis] = finite_diff_eps
            cols.append(f(x + e) - f(x - e))
        return torch.cat(cols, dim=-1) / (finite_diff_eps * 2.0)
    if method == 'autograd':
        assert x.requires_grad, "requires_grad need to be true for autograd!"
        y = f(x)
        from miso_amd import ops
        with ops.coordinate_gradient_only():      # (only d y / d x is asked for: no grid gradients on the way)
            return torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), create_graph=create_graph)[0]
    raise ValueError("Unknown method: {}".format(method))
