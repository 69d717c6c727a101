"""The brick push (miso_amd/csrc/grad_brick.hip, round 5): levels whose bricks are beyond the owner-computes pull are
accumulated per sort tile in LDS (double) and gathered by the owning tiles, instead of being scattered with float atomics
from the backward kernel.  Semantics = the grid half of grid_sampler_3d_backward (ATen; second-order sibling:
third_party/cuda_gridsample_grad2/gridsample_cuda.cu:462-481).  Checked against the unbinned atomic scatter of the same
library (itself pinned to the oracle in test_hip_parity.py / test_train_fused.py) and against the CPU oracle directly."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_torch as R  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(C, dims, bound, n, seed, crowd=0.5, ignore=None, H=64):
    from miso_amd import ops
    g = torch.Generator().manual_seed(seed)
    feats = [(torch.randn(1, C, z, y, x, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             for (x, y, z) in dims]
    F = C * len(dims)
    lin = [torch.nn.Linear(F, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    torch.manual_seed(seed)
    for m in lin:
        torch.nn.init.normal_(m.weight, std=0.3)
    pack = ops.DecoderPack([m.weight.detach().to(DEV) for m in lin], [m.bias.detach().to(DEV) for m in lin])
    meta = ops.GridMeta.from_bound(bound, ignore_level=ignore)
    b = torch.tensor(bound)
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.08 + b[:, 0] - 0.04 * (b[:, 1] - b[:, 0])
    k = int(n * crowd)
    if k:       # a crowd on a few tiles, straddling tile faces
        x[:k] = b.mean(1) + (torch.rand(k, 3, generator=g) - 0.5) * (b[:, 1] - b[:, 0]) * 0.11
    x[5] = float("nan")
    # samples exactly ON tile faces and on vertex positions (where the region's guard layer is needed)
    T = 16
    for a in range(3):
        t = torch.arange(0, T + 1, dtype=torch.float32)
        x[10 + a * 20: 10 + a * 20 + T + 1, a] = b[a, 0] + (b[a, 1] - b[a, 0]) * t / T
    return feats, meta, pack, x.to(DEV).contiguous(), lin


CASES = {
    # name: (C, level dims (x, y, z), bound)
    "scannet_small": (4, [(20, 10, 20), (150, 70, 160)], [[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]]),
    "scannet": (4, [(40, 20, 40), (200, 100, 200)], [[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]]),
    "odd_c8": (8, [(37, 50, 21), (147, 118, 93)], [[-1.0, 2.0], [0.0, 4.0], [-3.0, 0.5]]),
    "three": (4, [(16, 16, 16), (64, 64, 64), (176, 144, 160)], [[-1.0, 1.0], [-1.0, 1.0], [-1.0, 1.0]]),
}


@pytest.mark.parametrize("name", list(CASES))
def test_brick_levels_equal_the_atomic_scatter(name, monkeypatch):
    """sdf_bwd over a binned batch with the brick push against the unbinned atomic scatter: every level's gradient to the
    order of the float atomics (2e-5 of the largest entry), overwrite mode onto garbage (no zero-fill needed) and
    accumulate mode onto existing values, grad_touched flags == where the gradient is non-zero, and run to run the
    brick levels are reproducible up to the rounding of an fp64 sum to fp32."""
    from miso_amd import ops
    C, dims, bound = CASES[name]
    n = 180000 if name != "scannet" else 300000
    feats, meta, pack, x, _ = _case(C, dims, bound, n, seed=len(name))
    L = len(feats)
    gs = torch.randn(n, 1, device=DEV)
    sdf, mask = ops.sdf_fwd_raw(x, feats, meta, pack, True)
    _, ref = ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask, False, [True] * L)          # atomics, unbinned
    monkeypatch.setattr(ops.SortedBatch, "use_brick", True)
    sb = ops.SortedBatch(n, DEV).sort(x, meta)
    grads0 = [torch.empty_like(f) for f in feats]
    lv = int(ops._lib.load().miso_grad_brick_levels(ops.C.byref(ops._fill_grid(feats, meta, grads0, data=False)), sb.tiles, n))
    assert lv & (1 << (L - 1)), "the finest level is beyond the pull: it must go through the brick push"
    sdf_b, mask_b = ops.sdf_fwd_raw(x, feats, meta, pack, True, sorted_batch=sb)
    outs = []
    for rep in range(2):
        grads = [torch.full_like(f, 5.0 + rep) for f in feats]                            # garbage: overwrite must not care
        ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_b, False, [True] * L, grads, sorted_batch=sb, overwrite=True)
        outs.append([g.clone() for g in grads])
    assert sb.struct.brick_stage_floats > 0
    for l, (a, b, r) in enumerate(zip(outs[0], outs[1], ref)):
        if (lv >> l) & 1:
            # reproducible: the sums are formed in double, so the order of a tile's samples (it changes from run to run)
            # shows, if at all, in the last bit of a handful of entries -- where an fp64 sum sits on an fp32 rounding boundary
            d = (torch.nan_to_num(a) - torch.nan_to_num(b)).abs()
            assert d.max().item() <= 2e-7 * torch.nan_to_num(a).abs().max().item()
            assert (d > 0).float().mean().item() <= 1e-5
        fin = torch.isfinite(r)
        assert torch.equal(fin, torch.isfinite(a))
        assert (a[fin] - r[fin]).abs().max().item() <= 2e-5 * r[fin].abs().max().item()
    # accumulate mode: grad += on top of what is there
    base = [torch.randn_like(f) for f in feats]
    acc = [b.clone() for b in base]
    ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_b, False, [True] * L, acc, sorted_batch=sb, overwrite=False)
    for a, b, o in zip(acc, base, outs[0]):
        fin = torch.isfinite(o)
        assert (a[fin] - (b[fin] + o[fin])).abs().max().item() <= 1e-6 * max(1.0, o[fin].abs().max().item())
    # grad_touched: a flag per 64 floats of the gradient, set where something non-zero went in
    touched = [torch.zeros((f.numel() + 63) // 64, dtype=torch.uint8, device=DEV) for f in feats]
    grads = [torch.full_like(f, -1.0) for f in feats]
    ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_b, False, [True] * L, grads, sorted_batch=sb, overwrite=True,
                    touched=touched)
    for g, t in zip(grads, touched):
        flat = torch.nan_to_num(g.permute(0, 2, 3, 4, 1).reshape(-1), nan=1.0)
        pad = torch.zeros(t.numel() * 64, device=DEV)
        pad[: flat.numel()] = flat
        assert torch.equal(t != 0, pad.view(-1, 64).ne(0).any(dim=1))


def test_brick_push_with_an_ignored_level_and_per_axis_tiles(monkeypatch):
    """An ignored level that the brick push carries is written as zeros (overwrite) and left alone (accumulate); a
    per-axis binning (MISO_TILES_XYZ) with a level still beyond the pull goes through the same kernels."""
    from miso_amd import ops
    monkeypatch.setattr(ops.SortedBatch, "use_brick", True)
    C, dims, bound = 4, [(32, 16, 32), (150, 70, 160)], CASES["scannet_small"][2]      # (both levels fit the brick push)
    n = 120000
    feats, meta, pack, x, _ = _case(C, dims, bound, n, seed=3, ignore=[True, False])
    gs = torch.randn(n, 1, device=DEV)
    sdf, mask = ops.sdf_fwd_raw(x, feats, meta, pack, True)
    _, ref = ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask, False, [True, True])
    for tiles in (16, (12, 7, 20)):
        sb = ops.SortedBatch(n, DEV, tiles=tiles).sort(x, meta)
        _, mask_b = ops.sdf_fwd_raw(x, feats, meta, pack, True, sorted_batch=sb)
        grads = [torch.full_like(f, 9.0) for f in feats]
        ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_b, False, [True, True], grads, sorted_batch=sb, overwrite=True)
        assert sb.struct.brick_stage_floats > 0
        assert float(torch.nan_to_num(grads[0]).abs().max()) == 0.0
        fin = torch.isfinite(ref[1])
        assert (grads[1][fin] - ref[1][fin]).abs().max().item() <= 2e-5 * ref[1][fin].abs().max().item()


@pytest.mark.parametrize("lt", ["L1", "L2"])
def test_brick_training_step_vs_cpu_oracle(lt, monkeypatch):
    """The one-launch training step with the brick push (sort -> sdf_train_kernel -> brick accumulate -> brick gather) on a
    ScanNet-shaped grid against the CPU oracle: loss to 1e-6, every level's gradient to 2e-4 in the Euclidean norm."""
    from miso_amd import ops
    from miso_amd.step import MappingStep
    monkeypatch.setattr(ops.SortedBatch, "use_brick", True)
    C, dims, bound = CASES["scannet_small"]
    n = 100000
    feats, meta, pack, x, lin = _case(C, dims, bound, n, seed=17)
    x = torch.nan_to_num(x, nan=0.1)
    g = torch.Generator().manual_seed(1)
    target = (torch.randn(n, 1, generator=g) * 0.1).to(DEV)
    step = MappingStep(feats, meta, pack, n, lt, 1.0, 0.0, 0.0, sort=True, keep_sdf=False, use_graph=False)
    assert step._fused_train()
    step.set_batch(x, target)
    step.run(); step.run()
    torch.cuda.synchronize()
    assert step.sorted.struct.brick_stage_floats > 0
    fc = [f.detach().cpu().contiguous().clone().requires_grad_(True) for f in feats]
    ws = [m.weight.detach() for m in lin]
    bs = [m.bias.detach() for m in lin]
    pred = R.sdf_stock(fc, torch.tensor(bound), x.cpu(), ws, bs)
    loss = R.miso_loss_regression(pred, target.cpu(), None, None, lt)
    gref = torch.autograd.grad(loss, fc)
    assert abs(step.loss.sum().item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
    # (Euclidean norm to 2e-4; the max norm is looser only because a ReLU pre-activation within fp32 rounding of zero may
    # be gated differently by two implementations -- counted in test_config_shapes.py::..._are_relu_ties)
    for a, b in zip(step.grads, gref):
        a = a.cpu()
        assert ((a - b).double().norm() / b.double().norm()).item() <= 2e-4
        assert (a - b).abs().max().item() <= 1e-3 * b.abs().max().item()
