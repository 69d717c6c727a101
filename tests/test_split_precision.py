"""Admissibility of the split-precision decoder (round 6).

The fused kernels evaluate the decoder's products as bf16x3 pieces on v_mfma_f32_32x32x16_bf16 (miso_amd/csrc/
mlp_split.hpp) unless MISO_F_EXACT_F32 asks for the exact fp32 FMA chains of rounds 1-5.  `dtype` stays "f32" only if the
split form is as good an fp32 evaluation as the exact one.  The bar: against the FLOAT64 oracle, the split
kernel's max and mean error is at most 2x the exact-fp32 kernel's own error against float64 -- for the SDF, the
coordinate gradient (a per-point linear image of the d-feat rows) and the grid gradients -- on the decoder / grid shapes
of BASELINE cfg-1, 2, 3 and 5 (cfg-4, the alignment, has no decoder in its path).

Two kinds of inputs:
 * "dyadic": power-of-two grids on [-1,1]^3, points on a 2^-9 lattice and grid values with 8 significant bits, so that the
   trilinear encode is EXACT in fp32 and in fp64 alike: what is compared is the decoder arithmetic alone;
 * the configs' real shapes at uniform random points (the encode's own fp32 rounding is then common to both forms).
ReLU ties: a point whose pre-activation lies within rounding of zero may be gated differently by ANY two fp32
evaluations (tests/test_config_shapes.py::test_full_size_cfg2_gradient_outliers_are_relu_ties); such points (float64
census) carry a zero cotangent here.
Arithmetic compared: grid_opt/models/modules.py:11-32 (MLPNet), grid_opt/models/grid_net.py:306-325.
"""
import os

import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 65536
TIE = 1e-6

# (name, level sizes (Z,Y,X) or cells, C, H, bound, dyadic)
SHAPES = [
    ("cfg1_dyadic", [(64, 64, 64)], 4, 32, [[-1.0, 1.0]] * 3, True),
    ("cfg2_dyadic", [(32, 32, 32), (64, 64, 64), (128, 128, 128)], 8, 64, [[-1.0, 1.0]] * 3, True),
    ("cfg3_dyadic", [(32, 16, 32), (128, 64, 128)], 4, 64, [[-1.0, 1.0]] * 3, True),
    ("cfg5_dyadic", [(16, 16, 16), (32, 32, 32), (64, 64, 64), (128, 128, 128)], 8, 64, [[-1.0, 1.0]] * 3, True),
    ("cfg2", [(32, 32, 32), (64, 64, 64), (128, 128, 128)], 8, 64, [[-1.0, 1.0]] * 3, False),
    ("cfg3", [(40, 20, 40), (200, 100, 200)], 4, 64, [[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]], False),
    ("cfg5", [(10, 30, 30), (20, 60, 60), (40, 120, 120), (80, 240, 240)], 8, 64,
     [[-30.0, 30.0], [-30.0, 30.0], [-5.0, 15.0]], False),
]


def _inputs(levels, C, H, bound, dyadic, seed):
    gen = torch.Generator().manual_seed(seed)
    b = torch.tensor(bound, dtype=torch.float32)
    if dyadic:
        x = torch.randint(-511, 512, (N, 3), generator=gen).float() / 512.0
        feats = [torch.randint(-128, 128, (1, C) + tuple(s), generator=gen).float() / 4096.0 for s in levels]
    else:
        x = b[:, 0] + (b[:, 1] - b[:, 0]) * torch.rand(N, 3, generator=gen)
        feats = [torch.randn((1, C) + tuple(s), generator=gen) * 3e-2 for s in levels]
    F_ = C * len(levels)
    torch.manual_seed(seed)
    lin = [torch.nn.Linear(F_, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    ws = [l.weight.detach().clone() for l in lin]
    bs = [l.bias.detach().clone() for l in lin]
    ws[0] = ws[0] * 8.0          # pre-activations of order one from features of order 1e-2
    g = torch.randn(N, 1, generator=gen)
    return x, feats, b, ws, bs, g


def _oracle64(x, feats, b, ws, bs, g):
    xd = x.double().requires_grad_(True)
    fd = [f.double().requires_grad_(True) for f in feats]
    wd, bd = [w.double() for w in ws], [v.double() for v in bs]
    rows = R.encode_stock(fd, b.double(), xd)
    pre1 = rows @ wd[0].T + bd[0]
    pre2 = torch.relu(pre1) @ wd[1].T + bd[1]
    sdf = torch.relu(pre2) @ wd[2].T + bd[2]
    near = torch.minimum(pre1.detach().abs().min(dim=1).values, pre2.detach().abs().min(dim=1).values) < TIE
    g = g.clone()
    g[near] = 0.0
    got = torch.autograd.grad(sdf, [xd] + fd, g.double())
    return sdf.detach(), got[0], list(got[1:]), g, int(near.sum())


def _device(exact, x, feats, b, ws, bs, g):
    from miso_amd import ops
    meta = ops.GridMeta.from_bound(b)
    fd = [f.to(DEV).contiguous(memory_format=torch.channels_last_3d) for f in feats]
    pack = ops.DecoderPack([w.to(DEV) for w in ws], [v.to(DEV) for v in bs])
    xd = x.to(DEV)
    with ops.exact_fp32(exact):
        sdf, mask = ops.sdf_fwd_raw(xd, fd, meta, pack, want_mask=True)
        gx, grads = ops.sdf_bwd_raw(xd, fd, meta, pack, g.to(DEV), mask, True, [True] * len(fd))
    torch.cuda.synchronize()
    return sdf.cpu().double(), gx.cpu().double(), [t.cpu().double() for t in grads]


def _err(a, ref):
    d = (a - ref).abs()
    return d.max().item(), d.mean().item()


@pytest.mark.skipif(bool(os.environ.get("MISO_EXACT_F32")), reason="MISO_EXACT_F32 forces the exact chains: nothing to compare")
@pytest.mark.parametrize("name,levels,C,H,bound,dyadic", SHAPES, ids=[s[0] for s in SHAPES])
def test_split_decoder_error_within_twice_the_exact_fp32_kernels(name, levels, C, H, bound, dyadic):
    x, feats, b, ws, bs, g = _inputs(levels, C, H, bound, dyadic, seed=len(name) * 7 + C)
    sdf64, gx64, gf64, g, n_near = _oracle64(x, feats, b, ws, bs, g)
    assert n_near < 0.01 * N
    ex = _device(True, x, feats, b, ws, bs, g)
    sp = _device(False, x, feats, b, ws, bs, g)
    assert not torch.equal(ex[0], sp[0]), "the two decoder forms returned the same bits: is the switch connected?"
    report = []
    # a floor of a few ulps of the quantity's scale: where both errors are at the last bit, their ratio is noise
    for what, e, s, ref in [("sdf", ex[0], sp[0], sdf64), ("grad_x", ex[1], sp[1], gx64)] + \
            [(f"grad_level{l}", ex[2][l], sp[2][l], gf64[l]) for l in range(len(feats))]:
        (emax, emean), (smax, smean) = _err(e, ref), _err(s, ref)
        scale = ref.abs().max().item()
        report.append(f"{name} {what}: exact max {emax:.3e} mean {emean:.3e} | split max {smax:.3e} mean {smean:.3e} "
                      f"| scale {scale:.3e}")
        assert smax <= 2.0 * emax + 1e-7 * scale, report[-1]
        assert smean <= 2.0 * emean + 2e-9 * scale, report[-1]
    print("\n".join(report))
    if dyadic:
        # the encode is exact on these inputs: the SDF error IS the decoder's, and it is at the fp32 rounding level
        assert _err(sp[0], sdf64)[0] <= 2e-6 * max(1.0, sdf64.abs().max().item())


def test_exact_fp32_form_still_passes_its_parity_checks():
    """The exact chains stay in the library behind MISO_F_EXACT_F32: the checks the default (split) form passes in
    tests/test_train_fused.py and tests/test_hip_parity.py, re-run on the exact form."""
    from miso_amd import ops
    import test_train_fused as T
    with ops.exact_fp32():
        T.test_unbinned_train_kernel_equals_forward_plus_backward((8, (32, 64, 128), 64), 3000)
        T.test_unbinned_train_kernel_equals_forward_plus_backward((4, (16, 80), 64), 65)
        T.test_unbinned_train_kernel_equals_forward_plus_backward((4, (48,), 32), 3000)


@pytest.mark.parametrize("n", [3000, 70000])
def test_fused_double_backward_equals_the_torch_linear_chain(n):
    """create_graph=True through the fused decoder (round 6: ops._SdfFusedBackward -- the first backward keeps its d-feat
    rows, the second-order encode differentiates them) against the graph rebuilt from encode + torch.nn.functional.linear
    (rounds 1-5, MISO_BWD2_TORCH), on an eikonal + |sdf| loss (grid_opt/loss_isdf.py:96-152,367-377): loss, and the
    gradient of every level.  n = 70 000 takes the binned forward (its sign bits are in tile order: the first backward
    then runs in that order too, _SdfFusedBackward._rows); ReLU-tie points carry no weight."""
    from miso_amd import ops
    levels, C, H = [(16, 16, 16), (32, 32, 32), (64, 64, 64)], 8, 64
    gen = torch.Generator().manual_seed(n)
    b = torch.tensor([[-1.0, 1.0]] * 3)
    x = (torch.rand(n, 3, generator=gen) * 1.9 - 0.95).to(DEV)
    feats = [(torch.randn((1, C) + s, generator=gen) * 3e-2).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             .requires_grad_(True) for s in levels]
    torch.manual_seed(1)
    lin = [torch.nn.Linear(C * 3, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    ws, bs = [l.weight.detach().to(DEV) * (8.0 if i == 0 else 1.0) for i, l in enumerate(lin)], [l.bias.detach().to(DEV) for l in lin]
    pack = ops.DecoderPack(ws, bs)
    meta = ops.GridMeta.from_bound(b)
    with torch.no_grad():      # tie census (float64)
        rows = ops.encode(x, [f.detach() for f in feats], meta).double()
        pre1 = rows @ ws[0].double().T + bs[0].double()
        pre2 = torch.relu(pre1) @ ws[1].double().T + bs[1].double()
        keep = (torch.minimum(pre1.abs().min(1).values, pre2.abs().min(1).values) >= 1e-6).float()

    def run(torch_chain):
        ops._BWD2_TORCH = torch_chain
        try:
            xd = x.clone().requires_grad_(True)
            sdf = ops.sdf_fused(xd, feats, meta, pack)
            with ops.coordinate_gradient_only():
                g, = torch.autograd.grad(sdf, xd, torch.ones_like(sdf), create_graph=True)
            loss = (keep * (g.norm(dim=1) - 1) ** 2).mean() + (keep * sdf.abs().view(-1)).mean()
            gf = torch.autograd.grad(loss, feats)
            return loss.detach(), [t.detach() for t in gf], g.detach()
        finally:
            ops._BWD2_TORCH = False

    l1, g1, gx1 = run(False)
    l2, g2, gx2 = run(True)
    assert abs(l1.item() - l2.item()) <= 1e-5 * abs(l2.item())
    assert (keep[:, None] * (gx1 - gx2)).abs().max().item() <= 1e-4 * gx2.abs().max().item()      # (a tie point's own d sdf / d x may differ)
    for a, c in zip(g1, g2):
        assert (a - c).abs().max().item() <= 2e-4 * c.abs().max().item(), (a - c).abs().max().item() / c.abs().max().item()


def test_binned_double_backward_equals_the_caller_order_one():
    """A training-size batch is binned by the forward (SortedBatch.AUTO_MIN_POINTS) and the first backward under
    create_graph=True then runs in the binned order: d sdf / d x, the level gradients of an eikonal loss and the gradient
    with respect to the cotangent of sdf (rows are linear in it) equal those of the caller-order path."""
    from miso_amd import ops
    levels, C, H, n = [(16, 16, 16), (32, 32, 32), (64, 64, 64)], 8, 64, 70000
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(n, 3, generator=gen) * 1.9 - 0.95).to(DEV)
    feats = [(torch.randn((1, C) + s, generator=gen) * 3e-2).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             .requires_grad_(True) for s in levels]
    torch.manual_seed(2)
    lin = [torch.nn.Linear(C * 3, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.detach().to(DEV) for l in lin], [l.bias.detach().to(DEV) for l in lin])
    meta = ops.GridMeta.from_bound(torch.tensor([[-1.0, 1.0]] * 3))
    wgt = torch.rand(n, 1, generator=gen).to(DEV)

    def run(auto_min):
        saved = ops.SortedBatch.AUTO_MIN_POINTS
        ops.SortedBatch.AUTO_MIN_POINTS = auto_min
        try:
            xd = x.clone().requires_grad_(True)
            cot = wgt.clone().requires_grad_(True)
            sdf = ops.sdf_fused(xd, feats, meta, pack)
            g, = torch.autograd.grad(sdf, xd, cot, create_graph=True)
            loss = ((g.norm(dim=1) - 0.5) ** 2).mean()
            got = torch.autograd.grad(loss, feats + [cot])
            return g.detach(), [t.detach() for t in got]
        finally:
            ops.SortedBatch.AUTO_MIN_POINTS = saved

    g_b, d_b = run(65536)
    g_c, d_c = run(None)
    assert torch.equal(g_b, g_c)                     # per point the same launch arithmetic, whatever the order
    for a, c in zip(d_b, d_c):
        assert (a - c).abs().max().item() <= 1e-5 * c.abs().max().item() + 1e-12


def test_first_backward_rows_are_the_decoder_cotangent_of_the_feature_rows():
    """miso_sdf_bwd_rows: the d-feat rows it hands out equal d(sum g sdf) / d(feature rows) of the oracle's decoder on the
    oracle's feature rows (float64), its coordinate gradient and grid gradients equal miso_sdf_bwd's bit for bit / to the
    order of the float atomics."""
    from miso_amd import ops
    levels, C, H = [(16, 16, 16), (32, 32, 32), (64, 64, 64)], 8, 64
    x, feats, b, ws, bs, g = _inputs(levels, C, H, [[-1.0, 1.0]] * 3, False, seed=21)
    n = 20001
    x, g = x[:n], g[:n]
    meta = ops.GridMeta.from_bound(b)
    fd = [f.to(DEV).contiguous(memory_format=torch.channels_last_3d) for f in feats]
    pack = ops.DecoderPack([w.to(DEV) for w in ws], [v.to(DEV) for v in bs])
    xd, gd = x.to(DEV), g.to(DEV)
    sdf, mask = ops.sdf_fwd_raw(xd, fd, meta, pack, want_mask=True)
    gx1, gr1, rows = ops.sdf_bwd_rows_raw(xd, fd, meta, pack, gd, mask, True, [True] * 3)
    gx2, gr2 = ops.sdf_bwd_raw(xd, fd, meta, pack, gd, mask, True, [True] * 3)
    torch.cuda.synchronize()
    assert torch.equal(gx1, gx2)
    for a, c in zip(gr1, gr2):
        assert (a - c).abs().max().item() <= 2e-6 * c.abs().max().item()
    # oracle: decoder cotangent of the rows, float64, tie points excluded
    r64 = R.encode_stock([f.double() for f in feats], b.double(), x.double()).requires_grad_(True)
    wd, bd = [w.double() for w in ws], [v.double() for v in bs]
    pre1 = r64 @ wd[0].T + bd[0]
    pre2 = torch.relu(pre1) @ wd[1].T + bd[1]
    out = torch.relu(pre2) @ wd[2].T + bd[2]
    keep = torch.minimum(pre1.detach().abs().min(1).values, pre2.detach().abs().min(1).values) >= TIE
    ref, = torch.autograd.grad(out, r64, g.double())
    d = (rows.cpu().double() - ref)[keep].abs()
    assert d.max().item() <= 1e-5 * ref.abs().max().item(), d.max().item() / ref.abs().max().item()
