"""dev / bench extra: one eikonal step through the fused decoder at cfg-2 -- sdf = fused(x), g = d sdf / d x with
create_graph=True, loss = mean (|g| - 1)^2 + mean |sdf|, backward to the grids (grid_opt/loss_isdf.py:96-152,367-377).
MISO_BWD2_TORCH=1: the double backward rebuilt from encode + torch.nn.functional.linear (rounds 1-5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402


def build(dev="cuda:0", n=262144, L=3, C=8, H=64):
    torch.manual_seed(0)
    feats = [(torch.randn(1, C, s, s, s, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
             for s in (32, 64, 128)[:L]]
    meta = ops.GridMeta.from_bound([[-1., 1.]] * 3)
    lin = [torch.nn.Linear(L * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.detach().to(dev) for l in lin], [l.bias.detach().to(dev) for l in lin])
    x = torch.rand(n, 3, device=dev) * 2 - 1
    return feats, meta, pack, x


def step(feats, meta, pack, x):
    xd = x.clone().requires_grad_(True)
    sdf = ops.sdf_fused(xd, feats, meta, pack)
    with ops.coordinate_gradient_only():          # as grid_opt/diff.py / loss_isdf.py wrap their gradient calls
        g, = torch.autograd.grad(sdf.sum(), xd, create_graph=True)
    loss = ((g.norm(dim=1) - 1) ** 2).mean() + sdf.abs().mean()
    for f in feats:
        f.grad = None
    loss.backward()
    return loss


def timed(iters=10):
    args = build()
    for _ in range(20):      # (past allocator growth and the clock ramp)
        step(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    print("eikonal step through the fused decoder, 262144 points (%s): %.1f us" %
          ("torch linear chain" if ops._BWD2_TORCH else "fused double backward", timed()))
