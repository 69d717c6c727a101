"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
the golden vectors captured from the reference.

Tolerances (fp32; atomics make the scatter order non-deterministic, the
reference's own tests allow nondet_tol=1e-5, test3d.py:149):
  features  max|d| <= 1e-6 * max(1, |f|inf)
  sdf       mean|d| <= 1e-5 (BASELINE north-star), max|d| <= 1e-5
  gradients rel. error <= 1e-4 (+ small absolute floor)
"""
import math

import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def G(name):
    return np.load(gc.golden_path(name))


def to_dev(feats, layout):
    out = []
    for f in feats:
        f = f.to(DEV)
        if layout == "cl":
            f = f.contiguous(memory_format=torch.channels_last_3d)
        out.append(f)
    return out


def setup_case(name, layout="cl"):
    from miso_amd import ops
    case = gc.CASES[name]
    feats = [T(f) for f in gc.make_features(case)]
    bound = torch.tensor(case["bound"], dtype=torch.float32)
    ws, bs = R.decoder_params({k: T(v) for k, v in gc.make_decoder(case).items()})
    x = T(gc.make_points(case))
    meta = ops.GridMeta.from_bound(bound)
    fd = [f.requires_grad_(True) for f in to_dev(feats, layout)]
    pack = ops.DecoderPack([w.to(DEV) for w in ws], [b.to(DEV) for b in bs])
    return case, feats, bound, ws, bs, x, meta, fd, pack


def relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("name", ["small", "cfg1", "cfg2"])
@pytest.mark.parametrize("layout", ["cl", "ncdhw"])
def test_encode_forward_vs_golden(name, layout):
    from miso_amd import ops
    case, feats, bound, ws, bs, x, meta, fd, pack = setup_case(name, layout)
    out = ops.encode(x.to(DEV), fd, meta).cpu()
    g = G(name)
    torch.testing.assert_close(out.detach(), T(g["feats"]), rtol=0, atol=1e-6)
    # stability grids: C = 1 (generic scalar path)
    stab = to_dev([T(f) for f in gc.make_stability(case)], layout)
    s = ops.encode(x.to(DEV), stab, meta).cpu()
    torch.testing.assert_close(s, T(g["stab"]), rtol=1e-6, atol=2e-6)


@pytest.mark.parametrize("name", ["small", "cfg1"])
@pytest.mark.parametrize("layout", ["cl", "ncdhw"])
def test_encode_backward_vs_oracle(name, layout):
    from miso_amd import ops
    case, feats, bound, ws, bs, x, meta, fd, pack = setup_case(name, layout)
    torch.manual_seed(3)
    go = torch.randn(x.shape[0], case["fdim"] * case["n_levels"])
    xd = x.to(DEV).requires_grad_(True)
    out = ops.encode(xd, fd, meta)
    grads = torch.autograd.grad(out, fd + [xd], go.to(DEV))
    fc = [f.clone().requires_grad_(True) for f in feats]
    xc = x.clone().requires_grad_(True)
    ref = R.encode_stock(fc, bound, xc)
    rg = torch.autograd.grad(ref, fc + [xc], go)
    for a, b in zip(grads, rg):
        assert a.shape == b.shape
        assert relerr(a.cpu(), b) < 1e-4


@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("ac", [False, True])
def test_grid_sample_3d_dropin(pad, ac):
    """cuda_gridsample.grid_sample_3d call shape (cuda_gridsample.py:17-19), value,
    first and second derivatives, in- and out-of-bounds (test3d.py:37-72)."""
    from miso_amd import ops
    torch.manual_seed(0)
    inp = torch.randn(1, 3, 5, 6, 7)
    grid = torch.rand(1, 50, 1, 1, 3) * 2.6 - 1.3
    go = torch.randn(1, 3, 50, 1, 1)
    a_in = inp.to(DEV).requires_grad_(True)
    a_gr = grid.to(DEV).requires_grad_(True)
    out = ops.grid_sample_3d(a_in, a_gr, padding_mode=pad, align_corners=ac)
    b_in = inp.clone().requires_grad_(True)
    b_gr = grid.clone().requires_grad_(True)
    ref = torch.nn.functional.grid_sample(b_in, b_gr, mode="bilinear", padding_mode=pad, align_corners=ac)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)
    ga = torch.autograd.grad(out, [a_in, a_gr], go.to(DEV), create_graph=True)
    gb = torch.autograd.grad(ref, [b_in, b_gr], go)
    torch.testing.assert_close(ga[0].cpu(), gb[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ga[1].cpu(), gb[1], rtol=1e-4, atol=1e-5)
    # second order against the any-order restatement
    c_in = inp.clone().requires_grad_(True)
    c_gr = grid.clone().requires_grad_(True)
    r2 = R.trilinear_gather(c_in, c_gr.reshape(-1, 3), ac, pad)                   # (N,C)
    gc_ = torch.autograd.grad(r2, [c_in, c_gr], go[0, :, :, 0, 0].t(), create_graph=True)
    s_a = (ga[1] ** 2).sum() + (ga[0] ** 2).sum()
    s_c = (gc_[1] ** 2).sum() + (gc_[0] ** 2).sum()
    h_a = torch.autograd.grad(s_a, [a_in, a_gr])
    h_c = torch.autograd.grad(s_c, [c_in, c_gr])
    assert relerr(h_a[0].cpu(), h_c[0]) < 1e-4
    assert relerr(h_a[1].cpu(), h_c[1]) < 1e-4


def test_second_order_eikonal_vs_oracle():
    """L = mean((|grad_x sdf| - 1)^2) through encode + torch MLP: dL/dfeature, dL/dx
    against the fp32 any-order restatement."""
    from miso_amd import ops
    case, feats, bound, ws, bs, x, meta, fd, pack = setup_case("small")
    x = x[:256]

    def eik(enc, fs, xx, w, b):
        sdf = R.mlp_forward(enc(fs, xx), w, b)
        (gx,) = torch.autograd.grad(sdf, xx, torch.ones_like(sdf), create_graph=True)
        return ((gx.norm(dim=-1) - 1) ** 2).mean()

    xd = x.to(DEV).requires_grad_(True)
    la = eik(lambda fs, xx: ops.encode(xx, fs, meta), fd, xd, [w.to(DEV) for w in ws], [b.to(DEV) for b in bs])
    ga = torch.autograd.grad(la, fd + [xd])
    fc = [f.clone().requires_grad_(True) for f in feats]
    xc = x.clone().requires_grad_(True)
    lc = eik(lambda fs, xx: R.encode_gather(fs, bound, xx), fc, xc, ws, bs)
    gcpu = torch.autograd.grad(lc, fc + [xc])
    assert abs(la.item() - lc.item()) < 1e-5 * max(1.0, abs(lc.item()))
    for a, b in zip(ga, gcpu):
        assert relerr(a.cpu(), b) < 2e-4


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("name", ["small", "cfg1", "cfg2"])
def test_sdf_fused_vs_golden(name):
    from miso_amd import ops
    case, feats, bound, ws, bs, x, meta, fd, pack = setup_case(name)
    assert ops.sdf_fused_supported(fd, meta, pack)
    g = G(name)
    xd = x.to(DEV).requires_grad_(True)
    n = x.shape[0]
    sdf_t, valid, sign, weight = [T(a).to(DEV) for a in gc.make_targets(case, n)]
    pred = ops.sdf_fused(xd, fd, meta, pack)
    d = (pred.detach().cpu() - T(g["sdf"])).abs()
    assert d.mean().item() <= 1e-5 and d.max().item() <= 1e-5
    l1 = R.miso_loss_regression(pred, sdf_t, valid, weight, "L1")
    fs = R.miso_loss_free_space(pred, sdf_t, sign, 0.15)
    assert abs(l1.item() - float(g["loss_l1"])) < 1e-6
    grads = torch.autograd.grad(l1 + 0.1 * fs, fd + [xd])
    assert relerr(grads[-1].cpu(), T(g["grad_x"])) < 1e-4
    for l in range(case["n_levels"]):
        gl = grads[l].cpu().reshape(-1)
        idx = T(g[f"gfeat{l}_idx"])
        ref = T(g[f"gfeat{l}_val"])
        assert (gl[idx] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-10
        s = gl.double().abs().sum().item()
        assert abs(s - float(g[f"gfeat{l}_abssum"])) <= 1e-4 * float(g[f"gfeat{l}_abssum"])
    # smooth loss as well (better conditioned)
    pred2 = ops.sdf_fused(xd, fd, meta, pack)
    l2 = R.miso_loss_regression(pred2, sdf_t, valid, weight, "L2")
    g2 = torch.autograd.grad(l2, fd + [xd])
    assert relerr(g2[-1].cpu(), T(g["grad2_x"])) < 1e-4
    for l in range(case["n_levels"]):
        gl = g2[l].cpu().reshape(-1)
        ref = T(g[f"g2feat{l}_val"])
        assert (gl[T(g[f"gfeat{l}_idx"])] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-12


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000])
def test_ragged_sizes_and_ignore_level(n):
    from miso_amd import ops
    case, feats, bound, ws, bs, x, meta, fd, pack = setup_case("small")
    x = x[:n] if n <= x.shape[0] else torch.cat([x, x[: n - x.shape[0]]])
    meta_ig = ops.GridMeta.from_bound(bound, ignore_level=[True, False])
    for m, ig in ((meta, None), (meta_ig, [True, False])):
        xd = x.to(DEV).requires_grad_(True)
        out = ops.sdf_fused(xd, fd, m, pack)
        enc = ops.encode(xd, fd, m)
        xc = x.clone().requires_grad_(True)
        fc = [f.clone().requires_grad_(True) for f in feats]
        ref_e = R.encode_stock(fc, bound, xc, ig)
        ref = R.mlp_forward(ref_e, ws, bs)
        assert out.shape == (n, 1) and enc.shape == ref_e.shape
        if n == 0:
            continue
        torch.testing.assert_close(enc.detach().cpu(), ref_e.detach(), rtol=0, atol=1e-6)
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=0, atol=1e-5)
        ga = torch.autograd.grad(out.sum(), fd + [xd])
        gb = torch.autograd.grad(ref.sum(), fc + [xc], allow_unused=True)
        for a, b in zip(ga, gb):
            b = torch.zeros_like(a.cpu()) if b is None else b
            assert (a.cpu() - b).abs().max().item() <= 1e-4 * max(b.abs().max().item(), 1e-3)


def test_far_outside_and_nan_points_are_safe():
    from miso_amd import ops
    case, feats, bound, ws, bs, x, meta, fd, pack = setup_case("small")
    x = torch.tensor([[1e9, 0, 0], [-1e9, 1e9, 3e9], [float("nan"), 0.1, 0.2], [float("inf"), 0, 1.0],
                      [0.1, 0.1, 1.0]], dtype=torch.float32)
    enc = ops.encode(x.to(DEV), fd, meta).cpu()
    assert torch.all(enc[:2] == 0) and torch.all(enc[3] == 0)
    ref = R.encode_stock(feats, bound, x[4:5])
    torch.testing.assert_close(enc[4:5].detach(), ref, rtol=0, atol=1e-6)
    out = ops.sdf_fused(x.to(DEV), fd, meta, pack)
    out.sum().backward()
    torch.cuda.synchronize()
    for f in fd:
        assert torch.isfinite(f.grad).all()


# --------------------------------------------------------------------------- #
def test_full_size_properties_cfg2():
    """BASELINE cfg-2 full size (262 144 points): size-independent properties.
    (a) partition of unity on an all-ones grid, (b) linearity in the features,
    (c) adjointness <encode(G;x), r> == <G, encode_bwd(r;x)>, (d) the fused
    forward equals encode + torch MLP, (e) fused backward equals unfused backward."""
    from miso_amd import ops
    case = gc.CASES["cfg2"]
    n = 262144
    gen = torch.Generator().manual_seed(1234)
    x = (torch.rand(n, 3, generator=gen) * 2 - 1).to(DEV)
    bound = torch.tensor(case["bound"], dtype=torch.float32)
    meta = ops.GridMeta.from_bound(bound)
    shapes = [gc.grid_shape(case["bound"], c, case["fdim"]) for c in gc.level_cells(case)]
    g0 = torch.Generator(device=DEV).manual_seed(0)
    mk = lambda: [(torch.randn(s, device=DEV, generator=g0) * 1e-2).contiguous(
        memory_format=torch.channels_last_3d) for s in shapes]
    A, B = mk(), mk()
    ones = [torch.ones_like(a) for a in A]
    e1 = ops.encode(x, ones, meta)
    # points at least half a COARSEST cell (1/32 in normalised units) inside the bound see
    # all 8 corners at every level
    inner = (x.abs() < 1 - 1.0 / 32 - 1e-4).all(dim=1)
    assert (e1[inner] - 1).abs().max().item() < 2e-6
    ea, eb = ops.encode(x, A, meta), ops.encode(x, B, meta)
    comb = ops.encode(x, [2.5 * a - 0.75 * b for a, b in zip(A, B)], meta)
    assert (comb - (2.5 * ea - 0.75 * eb)).abs().max().item() < 1e-6
    r = torch.randn(n, ea.shape[1], device=DEV, generator=g0)
    Ag = [a.requires_grad_(True) for a in A]
    ea = ops.encode(x, Ag, meta)
    grads = torch.autograd.grad(ea, Ag, r)
    lhs = (ea.detach().double() * r.double()).sum().item()
    rhs = sum((a.detach().double() * g.double()).sum().item() for a, g in zip(Ag, grads))
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    # fused vs unfused
    sd = {k: T(v).to(DEV) for k, v in gc.make_decoder(case).items()}
    ws, bs = R.decoder_params(sd)
    pack = ops.DecoderPack(ws, bs)
    fused = ops.sdf_fused(x, Ag, meta, pack)
    unf = R.mlp_forward(ops.encode(x, Ag, meta), ws, bs)
    d = (fused - unf).abs()
    assert d.mean().item() <= 1e-6 and d.max().item() <= 1e-5
    go = torch.randn(n, 1, device=DEV, generator=g0) / n
    # ReLU ties (as tests/test_config_shapes.py::test_full_size_cfg2_gradient_outliers_are_relu_ties): two fp32 evaluations
    # of the decoder -- rocBLAS here, the kernel's matrix-core products there (bf16x3 pieces since round 6) -- may gate a
    # ReLU differently where its pre-activation lies within rounding of zero, and such a point moves the gradient of the
    # fine-level vertices it touches by O(1 %).  Those points (float64 census, a handful of 262 144) carry no cotangent.
    with torch.no_grad():
        rows64 = ops.encode(x, [a.detach() for a in Ag], meta).double()
        pre1 = rows64 @ ws[0].double().T + bs[0].double()
        pre2 = torch.relu(pre1) @ ws[1].double().T + bs[1].double()
        near = torch.minimum(pre1.abs().min(dim=1).values, pre2.abs().min(dim=1).values) < 2e-7
    assert int(near.sum()) < 3e-4 * n
    go[near] = 0.0
    gf = torch.autograd.grad(fused, Ag, go)
    gu = torch.autograd.grad(unf, Ag, go)
    for a, b in zip(gf, gu):
        assert relerr(a, b) < 1e-4


def test_adam_dense_vs_torch():
    from miso_amd import ops
    torch.manual_seed(0)
    p0 = torch.randn(1, 4, 9, 7, 5)
    grads = [torch.randn_like(p0) * (i + 1) * 1e-3 for i in range(4)]
    pc = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pc], lr=1e-3)
    for g in grads:
        pc.grad = g.clone()
        opt.step()
    for fmt in (torch.contiguous_format, torch.channels_last_3d):
        p = p0.to(DEV).contiguous(memory_format=fmt)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for i, g in enumerate(grads):
            gd = torch.empty_like(p).copy_(g.to(DEV))
            ops.adam_dense_(p, gd, m, v, i + 1, 1e-3, zero_grad=(i % 2 == 0))
            assert (gd == 0).all() == (i % 2 == 0)
        torch.testing.assert_close(p.cpu(), pc.detach(), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(m.cpu(), opt.state[pc]["exp_avg"], rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(v.cpu(), opt.state[pc]["exp_avg_sq"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("numel", [256 * 40, 256 * 37 + 130, 1027, 5])
def test_adam_active_is_bit_identical_to_dense(numel):
    """miso_adam_active steps only the chunks a gradient has ever reached; everything -- parameters, both moments,
    cleared gradients -- must equal the dense kernel bit for bit over a run in which chunks wake up at different
    steps, some never, and one gradient is NaN.  Also checked against torch.optim.Adam."""
    from miso_amd import ops, _lib
    g = torch.Generator().manual_seed(numel)
    p0 = torch.randn(numel, generator=g)
    nchunks = (numel + _lib.ADAM_CHUNK - 1) // _lib.ADAM_CHUNK
    wake = torch.randint(0, 8, (nchunks,), generator=g)         # step at which a chunk first gets a gradient; >= 6: never
    grads = []
    for t in range(6):
        gr = torch.randn(numel, generator=g) * 1e-2
        on = ((wake <= t) & (torch.rand(nchunks, generator=g) < 0.7)).repeat_interleave(_lib.ADAM_CHUNK)[:numel]
        gr = gr * on * (torch.rand(numel, generator=g) < 0.3)      # sparse inside a chunk as well
        grads.append(gr)
    if numel > 300:
        grads[2][7] = float("nan")
    res = {}
    for kind in ("dense", "active"):
        p, m, v = p0.to(DEV).clone(), torch.zeros(numel, device=DEV), torch.zeros(numel, device=DEV)
        act = ops.adam_active_flags(p)
        for t, gr in enumerate(grads):
            gd = gr.to(DEV).clone()
            zero = t % 2 == 1
            if kind == "dense":
                ops.adam_dense_(p, gd, m, v, t + 1, 1e-2, zero_grad=zero)
            else:
                ops.adam_active_(p, gd, m, v, act, t + 1, 1e-2, zero_grad=zero)
            assert bool((gd == 0).all()) == zero or not zero
            if zero:
                assert bool((gd == 0).all())
        res[kind] = (p, m, v, act)
    for a, b in zip(res["dense"][:3], res["active"][:3]):
        assert torch.equal(a.cpu().view(torch.int32), b.cpu().view(torch.int32))
    ever = torch.stack([(gr != 0) for gr in grads]).any(0)
    ever = torch.nn.functional.pad(ever, (0, nchunks * _lib.ADAM_CHUNK - numel)).reshape(nchunks, -1).any(1)
    assert torch.equal(res["active"][3].cpu().bool(), ever)
    if numel <= 300:                                            # no NaN in this run: compare with torch's Adam
        pc = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([pc], lr=1e-2)
        for gr in grads:
            pc.grad = gr.clone()
            opt.step()
        torch.testing.assert_close(res["active"][0].cpu(), pc.detach(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("numel", [256 * 40, 256 * 37 + 130, 1027, 5, 64 * 3, 64 * 4 + 1, 256 * 3 + 64 * 2 + 7,
                                   256 * 8 * 16 * 5 + 64 * 3 + 9])
def test_adam_touched_ragged_sizes_equal_the_gradient_scan(numel):
    """miso_adam_touched on sizes that end inside a slab / inside a chunk / inside a flag word, flags written by hand
    (touched = the chunks of this step's non-zero gradients): parameters, moments, `active`, cleared gradients and
    cleared flags equal miso_adam_active's bit for bit over a run with chunks waking up at different steps and a NaN
    guard."""
    from miso_amd import ops, _lib
    CH = _lib.ADAM_CHUNK
    g = torch.Generator().manual_seed(numel + 1)
    p0 = torch.randn(numel, generator=g)
    nchunks = (numel + CH - 1) // CH
    wake = torch.randint(0, 7, (nchunks,), generator=g)
    res = {}
    for kind in ("scan", "flags"):
        gg = torch.Generator().manual_seed(numel + 2)
        p, m, v = p0.to(DEV).clone(), torch.zeros(numel, device=DEV), torch.zeros(numel, device=DEV)
        act = ops.adam_active_flags(p)
        tch = ops.adam_active_flags(p)
        for t in range(5):
            gr = torch.randn(numel, generator=gg) * 1e-2
            on = ((wake <= t) & (torch.rand(nchunks, generator=gg) < 0.6)).repeat_interleave(CH)[:numel]
            gr = gr * on * (torch.rand(numel, generator=gg) < 0.4)
            gd = gr.to(DEV).clone()
            guard = torch.tensor([float("nan") if t == 3 else 0.5], device=DEV)
            zero = t % 2 == 0
            if kind == "scan":
                ops.adam_active_(p, gd, m, v, act, t + 1, 1e-2, zero_grad=zero, guard=guard)
            else:
                nz = torch.nn.functional.pad(gr != 0, (0, nchunks * CH - numel)).reshape(nchunks, CH).any(1)
                tch.copy_(nz.to(torch.uint8))
                ops.adam_active_(p, gd, m, v, act, t + 1, 1e-2, zero_grad=zero, guard=guard, touched=tch)
                assert int(tch.sum()) == 0
            res.setdefault(kind, []).append((p.clone(), m.clone(), v.clone(), act.clone(), gd.clone()))
    for a, b in zip(res["scan"], res["flags"]):
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x,
                               y.view(torch.int32) if y.dtype == torch.float32 else y)


def test_native_library_is_loaded():
    """The ops above ran through libmiso_hip.so (no eager fallback exists)."""
    from miso_amd import _lib
    assert _lib._lib is not None
    maps = open("/proc/self/maps").read()
    assert "libmiso_hip.so" in maps
    # ... and that library was built from the sources it travels with
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("srchash", os.path.join(root, "miso_amd", "csrc", "srchash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert _lib.load().miso_version().decode().endswith("src=" + mod.source_hash()), "stale libmiso_hip.so"


# --------------------------------------------------------------------------- #
# spatially binned (sorted) path
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("tiles", [16, (25, 13, 25), (32, 32, 32), (5, 1, 9)])
@pytest.mark.parametrize("n", [1, 777, 40000, 300001])
def test_sort_points_is_a_tile_grouped_permutation(n, tiles):
    """tiles: a count (cubic binning) or per-axis counts packed as MISO_TILES_XYZ (include/miso_hip.h)."""
    from miso_amd import ops
    tx, ty, tz = (tiles,) * 3 if isinstance(tiles, int) else tiles
    bound = [[-1.0, 1.3], [-0.7, 0.9], [0.0, 2.1]]
    meta = ops.GridMeta.from_bound(bound)
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, 3, generator=g) * torch.tensor([2.5, 1.8, 2.3]) + torch.tensor([-1.1, -0.8, -0.1])
    x[0] = float("nan")
    sb = ops.SortedBatch(n, DEV, tiles=tiles, keep_metric=True).sort(x.to(DEV), meta)
    perm = sb.perm.cpu().long()
    assert torch.equal(torch.sort(perm).values, torch.arange(n))
    xs = sb.x_sorted.cpu()
    assert torch.equal(torch.nan_to_num(xs, nan=7.0), torch.nan_to_num(x[perm], nan=7.0))
    # the normalised float4 copy the kernels read: 2 (x - min) / len - 1, op for op
    b_ = torch.tensor(bound)
    xn = (2.0 * (xs - b_[:, 0])) / (b_[:, 1] - b_[:, 0]) - 1.0
    assert torch.equal(torch.nan_to_num(sb.xn_sorted.cpu()[:, :3], nan=7.0), torch.nan_to_num(xn, nan=7.0))
    # metric copy is optional: same permutation class without it
    sb2 = ops.SortedBatch(n, DEV, tiles=tiles).sort(x.to(DEV), meta)
    assert torch.equal(sb2.tile_offsets.cpu(), sb.tile_offsets.cpu())
    assert torch.equal(torch.sort(sb2.perm.cpu().long()).values, torch.arange(n))
    off = sb.tile_offsets.cpu().long()
    assert off[0] == 0 and off[-1] == n and torch.all(off[1:] >= off[:-1])
    b = torch.tensor(bound)
    u = torch.nan_to_num((xs - b[:, 0]) / (b[:, 1] - b[:, 0]), nan=0.0)
    tt = torch.tensor([tx, ty, tz])
    t = torch.minimum(torch.clamp(torch.floor(u * tt), min=0), (tt - 1).float()).long()
    tid = (t[:, 2] * ty + t[:, 1]) * tx + t[:, 0]
    expect = torch.repeat_interleave(torch.arange(tx * ty * tz), off[1:] - off[:-1])
    assert torch.equal(tid, expect)


def _check_sorted_batch(sb, x, bound, tiles3):
    n = x.shape[0]
    tx, ty, tz = tiles3
    perm = sb.perm.cpu().long()
    assert torch.equal(torch.sort(perm).values, torch.arange(n))
    assert torch.equal(sb.xn_sorted.cpu()[:, 3].view(torch.int32).long(), perm)      # the index also rides in .w
    off = sb.tile_offsets.cpu().long()
    assert off[0] == 0 and off[-1] == n and torch.all(off[1:] >= off[:-1])
    b = torch.tensor(bound)
    u = torch.nan_to_num((x[perm] - b[:, 0]) / (b[:, 1] - b[:, 0]), nan=0.0)
    tt = torch.tensor([tx, ty, tz])
    t = torch.minimum(torch.clamp(torch.floor(u * tt), min=0), (tt - 1).float()).long()
    tid = (t[:, 2] * ty + t[:, 1]) * tx + t[:, 0]
    assert torch.equal(tid, torch.repeat_interleave(torch.arange(tx * ty * tz), off[1:] - off[:-1]))


@pytest.mark.gpu
def test_sort_reuses_its_buffers_batch_after_batch():
    """The SAME SortedBatch (workspace, outputs) sorts batch after batch -- uniform, everything in one tile, a thin slab
    (most tiles empty), more than a million points -- and every result is a tile-grouped permutation with the original
    index in xn_sorted[:, 3]: nothing of one call leaks into the next."""
    from miso_amd import ops
    bound = [[-1.0, 1.0], [-2.0, 2.0], [0.0, 3.0]]
    meta = ops.GridMeta.from_bound(bound)
    lo, ln = torch.tensor([-1.0, -2.0, 0.0]), torch.tensor([2.0, 4.0, 3.0])
    g = torch.Generator().manual_seed(11)
    for n, tiles in [(70001, 16), (3000, (25, 16, 25)), ((1 << 20) + 4097, 16)]:
        sb = ops.SortedBatch(n, DEV, tiles=tiles)
        t3 = (tiles,) * 3 if isinstance(tiles, int) else tiles
        for kind in ("uniform", "one_tile", "uniform", "slab", "uniform"):
            x = torch.rand(n, 3, generator=g) * ln + lo
            if kind == "one_tile":
                x = x * 1e-3 + torch.tensor([0.3, 0.3, 1.1])
            if kind == "slab":
                x[:, 2] = x[:, 2] * 0.01 + 2.0
            sb.sort(x.to(DEV), meta)
            _check_sorted_batch(sb, x, bound, t3)
    sb0 = ops.SortedBatch(0, DEV)
    sb0.sort(torch.empty(0, 3, device=DEV), meta)
    assert int(sb0.tile_offsets.cpu().abs().sum()) == 0


@pytest.mark.parametrize("name,n", [("small", 5000), ("cfg2", 50000)])
def test_sorted_path_matches_unsorted(name, n):
    """Binned forward/backward (LDS pre-reduction per tile) == plain path up to fp32
    summation order, including points outside the bound and ragged tile sizes."""
    from miso_amd import ops
    case, feats, bound, ws, bs, x0, meta, fd, pack = setup_case(name)
    g = torch.Generator().manual_seed(5)
    b = torch.tensor(case["bound"])
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.1 + b[:, 0] - 0.05 * (b[:, 1] - b[:, 0])
    x[: n // 3] = x[: n // 3] * 0.05 + b.mean(dim=1)      # a dense cluster: very uneven tiles
    x = x.to(DEV)
    gs = torch.randn(n, 1, generator=g).to(DEV)
    L = len(fd)
    sdf_a, mask_a = ops.sdf_fwd_raw(x, fd, meta, pack, True)
    gx_a, gr_a = ops.sdf_bwd_raw(x, fd, meta, pack, gs, mask_a, True, [True] * L)
    sb = ops.SortedBatch(n, DEV).sort(x, meta)
    sdf_b, mask_b = ops.sdf_fwd_raw(x, fd, meta, pack, True, sorted_batch=sb)
    gx_b, gr_b = ops.sdf_bwd_raw(x, fd, meta, pack, gs, mask_b, True, [True] * L, sorted_batch=sb)
    assert torch.equal(sdf_a, sdf_b) or (sdf_a - sdf_b).abs().max().item() < 1e-7
    assert relerr(gx_b, gx_a) < 1e-5
    for a, b_ in zip(gr_a, gr_b):
        assert relerr(b_, a) < 2e-5
    # grid-only variant (the mapping step) through autograd, which bins automatically
    if n >= 40000:
        old = ops.SortedBatch.AUTO_MIN_POINTS
        ops.SortedBatch.AUTO_MIN_POINTS = 1
        try:
            out = ops.sdf_fused(x, fd, meta, pack)
            gg = torch.autograd.grad(out, fd, gs)
        finally:
            ops.SortedBatch.AUTO_MIN_POINTS = old
        for a, b_ in zip(gr_a, gg):
            assert relerr(b_, a) < 2e-5


def test_mapping_loss_kernel_vs_oracle():
    from miso_amd import ops
    case = gc.CASES["small"]
    n = 5001
    sdf_t, valid, sign, weight = [T(a) for a in gc.make_targets(case, n)]
    torch.manual_seed(0)
    pred = (torch.randn(n, 1) * 0.2)
    for lt in ("L1", "L2"):
        pc = pred.clone().requires_grad_(True)
        ref = 1.3 * R.miso_loss_regression(pc, sdf_t, valid, weight, lt) + \
            0.1 * R.miso_loss_free_space(pc, sdf_t, sign, 0.15)
        ref.backward()
        pd = pred.to(DEV).requires_grad_(True)
        out = ops.mapping_loss(pd, sdf_t.to(DEV), valid.to(DEV), sign.to(DEV), weight.to(DEV), lt, 1.3, 0.1, 0.15)
        out.sum().backward()
        assert abs(out.sum().item() - ref.item()) < 1e-6
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("binned,lt", [(False, "L1"), (True, "L1"), (True, "L2")])
def test_mapping_step_matches_autograd_and_adam(binned, lt):
    """MappingStep (captured launch sequence) == autograd through the same ops, and with
    Adam == torch.optim.Adam on the CPU oracle after 3 iterations.  binned: sort + forward +
    backward with the loss folded in (miso_sdf_bwd_sorted_loss) + pull."""
    from miso_amd import ops
    from miso_amd.step import MappingStep
    case, feats, bound, ws, bs, x0, meta, fd, pack = setup_case("small")
    n = 40000
    g = torch.Generator().manual_seed(9)
    b = torch.tensor(case["bound"])
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) + b[:, 0]
    sdf_t, valid, sign, weight = [T(a) for a in gc.make_targets(case, n)]
    params = [f.detach().clone() for f in fd]
    step = MappingStep(params, meta, pack, n, lt, 1.0, 0.1, 0.15, adam=dict(lr=1e-3), use_graph=False, sort=binned)
    step.set_batch(x.to(DEV), sdf_t.to(DEV), valid.to(DEV), sign.to(DEV), weight.to(DEV))
    fc = [torch.nn.Parameter(f.clone()) for f in feats]
    opt = torch.optim.Adam(fc, lr=1e-3)
    for it in range(3):
        step.run()
        opt.zero_grad()
        pred = R.sdf_stock(fc, bound, x, ws, bs)
        loss = R.miso_loss_regression(pred, sdf_t, valid, weight, lt) + \
            0.1 * R.miso_loss_free_space(pred, sdf_t, sign, 0.15)
        loss.backward()
        opt.step()
        assert abs(step.loss.sum().item() - loss.item()) < 1e-6 + 1e-5 * abs(loss.item())
    for a, b_ in zip(step.features, fc):
        assert (a.cpu() - b_.detach()).abs().max().item() < 2e-6
    # graph-captured variant without Adam gives the same gradients as eager
    s2 = MappingStep([f.detach() for f in fd], meta, pack, n, lt, 1.0, 0.1, 0.15, use_graph=True, sort=True)
    s2.set_batch(x.to(DEV), sdf_t.to(DEV), valid.to(DEV), sign.to(DEV), weight.to(DEV))
    s2.run(); s2.run()
    torch.cuda.synchronize()
    out = ops.sdf_fused(x.to(DEV), fd, meta, pack)
    l = ops.mapping_loss(out, sdf_t.to(DEV), valid.to(DEV), sign.to(DEV), weight.to(DEV), lt, 1.0, 0.1, 0.15).sum()
    gg = torch.autograd.grad(l, fd)
    for a, b_ in zip(s2.grads, gg):
        assert relerr(a, b_) < 2e-5
    assert abs(s2.loss.sum().item() - l.item()) < 1e-6 + 1e-5 * abs(l.item())


@pytest.mark.parametrize("shape", ["scannet", "four_level", "tiny_grid"])
def test_binned_backward_other_shapes(shape):
    """Binned (pull) backward == plain atomic backward on shapes that exercise the fallbacks:
    a level too fine for the pull (ScanNet's 200x100x200 at 16 tiles -> 13 vertices per tile:
    stays on the atomic scatter, cleared by the library), C = 4 (two levels share the MFMA
    register halves), four levels, grids smaller than the tile count (tiles owning no vertex)."""
    from miso_amd import ops
    torch.manual_seed(0)
    if shape == "scannet":
        bound, dims, C, H, n = [[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]], [(40, 20, 40), (200, 100, 200)], 4, 64, 70000
    elif shape == "four_level":
        bound, dims, C, H, n = [[-1.0, 1.0]] * 3, [(8, 8, 8), (16, 16, 16), (32, 32, 32), (64, 64, 64)], 8, 64, 70000
    else:
        bound, dims, C, H, n = [[0.0, 1.0], [0.0, 2.0], [-1.0, 0.0]], [(3, 5, 2), (7, 9, 5)], 4, 32, 66000
    feats = [(torch.randn(1, C, z, y, x_, device=DEV) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
             for (x_, y, z) in dims]
    F = C * len(dims)
    lin = [torch.nn.Linear(F, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.detach().to(DEV) for l in lin], [l.bias.detach().to(DEV) for l in lin])
    meta = ops.GridMeta.from_bound(bound)
    assert ops.sdf_fused_supported(feats, meta, pack)
    b = torch.tensor(bound)
    x = (torch.rand(n, 3) * (b[:, 1] - b[:, 0]) * 1.04 + b[:, 0] - 0.02 * (b[:, 1] - b[:, 0])).to(DEV)
    gs = torch.randn(n, 1, device=DEV)
    L = len(feats)
    sdf_a, mask_a = ops.sdf_fwd_raw(x, feats, meta, pack, True)
    gx_a, gr_a = ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_a, True, [True] * L)
    sb = ops.SortedBatch(n, DEV).sort(x, meta)
    sdf_b, mask_b = ops.sdf_fwd_raw(x, feats, meta, pack, True, sorted_batch=sb)
    # poisoned gradient buffers: overwrite mode must not depend on their content
    grads = [torch.full_like(f, float("nan")) for f in feats]
    gx_b, gr_b = ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_b, True, [True] * L, grads, sorted_batch=sb,
                                 overwrite=True)
    assert (sdf_a - sdf_b).abs().max().item() < 1e-7
    assert relerr(gx_b, gx_a) < 1e-5
    for a, b_ in zip(gr_a, gr_b):
        assert torch.isfinite(b_).all()
        assert relerr(b_, a) < 3e-5
    # accumulate mode (+=) on top of an existing gradient
    base = [torch.randn_like(f) for f in feats]
    acc = [t.clone() for t in base]
    ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_b, False, [True] * L, acc, sorted_batch=sb, overwrite=False)
    for a, b0, r in zip(acc, base, gr_a):
        assert relerr(a - b0, r) < 3e-4


@pytest.mark.parametrize("name,n", [("cfg1", 20000), ("cfg2", 70001)])
def test_encode_backward_pull_path_vs_oracle(name, n, monkeypatch):
    """Large batches take sort + owner-computes pull for the grid half of the encode backward
    (ops.ENCODE_PULL_MIN_POINTS); same result as the atomic scatter and as the CPU oracle,
    with points outside the bound, a strided grad_output and an ignored level."""
    from miso_amd import ops
    case, feats, bound, ws, bs, x0, meta, fd, pack = setup_case(name)
    L, F = case["n_levels"], case["fdim"] * case["n_levels"]
    g = torch.Generator().manual_seed(11)
    b = torch.tensor(case["bound"])
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.1 + b[:, 0] - 0.05 * (b[:, 1] - b[:, 0])
    go_wide = torch.randn(n, F + 8, generator=g).to(DEV)
    go = go_wide[:, 4:4 + F]                      # row pitch F + 8, 16-B aligned base
    assert ops.ENCODE_PULL_MIN_POINTS is not None and n >= ops.ENCODE_PULL_MIN_POINTS

    def run():
        xd = x.to(DEV).requires_grad_(True)
        out = ops.encode(xd, fd, meta)
        return torch.autograd.grad(out, fd + [xd], go)

    got = run()
    out_binned = ops.encode(x.to(DEV), fd, meta).detach()       # binned forward (features need grad)
    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", None)
    atomic = run()
    assert torch.equal(out_binned, ops.encode(x.to(DEV), fd, meta).detach())   # same op order per point
    for a, c in zip(got, atomic):
        assert relerr(a, c) < 2e-5
    fc = [f.clone().requires_grad_(True) for f in feats]
    xc = x.clone().requires_grad_(True)
    ref = torch.autograd.grad(R.encode_stock(fc, bound, xc), fc + [xc], go.cpu())
    for a, c in zip(got, ref):
        assert a.shape == c.shape and relerr(a.cpu(), c) < 1e-4
    # an ignored level gets an exactly zero gradient from the pull as well
    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", 16384)
    meta_ig = ops.GridMeta.from_bound(bound, ignore_level=[l == L - 1 for l in range(L)])
    gi = torch.autograd.grad(ops.encode(x.to(DEV), fd, meta_ig), fd, go, allow_unused=True)
    assert gi[L - 1] is None or float(gi[L - 1].abs().max()) == 0.0
    if L > 1:
        assert relerr(gi[0], got[0]) < 2e-5


@pytest.mark.parametrize("lt", ["L2", "GM"])
@pytest.mark.parametrize("n", [1, 1000, 16384])
def test_lm_normal_equations_vs_oracle(lt, n):
    """miso_lm_normal_eq == the tracker's J^T W J / J^T W r built op by op (tracker.py:176-196)."""
    from miso_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 3, generator=g)
    gw = torch.randn(n, 3, generator=g)
    s = torch.randn(n, 1, generator=g) * 0.2
    t = torch.randn(n, 1, generator=g) * 0.2
    Rm = torch.tensor(gc.rodrigues(np.array([0.3, -0.2, 0.5])), dtype=torch.float32)
    _, H_ref, g_ref = R.lm_normal_equations(x.double(), Rm.double(), gw.double(), s.double(), t.double(), lt, 0.1)
    H, gv, wr2 = ops.lm_normal_eq(x.to(DEV), Rm.to(DEV), gw.to(DEV), s.to(DEV), t.to(DEV), lt, 0.1)
    scale = max(H_ref.abs().max().item(), 1e-30)
    assert (H.cpu().double() - H_ref).abs().max().item() < 2e-5 * scale
    assert (gv.cpu().double() - g_ref).abs().max().item() < 2e-5 * max(g_ref.abs().max().item(), 1e-30)
    assert torch.equal(H, H.T)


@pytest.mark.parametrize("n", [0, 5, 100003])
def test_overlap_count_vs_oracle(n):
    """miso_overlap_count == the reference's transform -> transform -> coords_in_bound -> count
    (grid_atlas.py:405-420) on the CPU oracle."""
    from miso_amd import ops
    g = torch.Generator().manual_seed(n + 1)
    pts = torch.rand(n, 3, generator=g) * 6 - 3
    R_s = torch.tensor(gc.rodrigues(np.array([0.2, -0.1, 0.4])), dtype=torch.float32)
    R_d = torch.tensor(gc.rodrigues(np.array([-0.3, 0.25, 0.1])), dtype=torch.float32)
    t_s, t_d = torch.tensor([[0.3], [-0.2], [0.1]]), torch.tensor([[-0.4], [0.5], [0.2]])
    bound = torch.tensor([[-1.0, 1.5], [-2.0, 0.7], [-0.5, 2.2]])
    world = R.transform_points_to(pts, R_s, t_s)
    local = R.transfrom_points_from(world, R_d, t_d)
    ref = int(torch.count_nonzero(R.coords_in_bound(local, bound)))
    got = ops.overlap_count(R_s.to(DEV), t_s.to(DEV), R_d.to(DEV), t_d.to(DEV), pts.to(DEV), bound)
    # a point within a few ulps of a face may fall on either side (different summation order)
    assert abs(int(got.item()) - ref) <= max(2, n // 50000)


@pytest.mark.parametrize("name,n", [("small", 20000), ("cfg2", 40000)])
def test_second_order_pull_path_vs_atomic_and_oracle(name, n, monkeypatch):
    """Eikonal-type loss on a batch large enough for the binned path: the grid gradient of the
    SECOND backward comes from miso_grad_pull_dx; same as the atomic scatter of miso_encode_bwd2
    and (on a subset small enough for the CPU) as the any-order oracle."""
    from miso_amd import ops
    case, feats, bound, ws, bs, x0, meta, fd, pack = setup_case(name)
    g = torch.Generator().manual_seed(21)
    b = torch.tensor(case["bound"])
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.1 + b[:, 0] - 0.05 * (b[:, 1] - b[:, 0])
    wd, bd = [w.to(DEV) for w in ws], [bb.to(DEV) for bb in bs]

    def eik(enc, fs, xx, w, b_):
        sdf = R.mlp_forward(enc(fs, xx), w, b_)
        (gx,) = torch.autograd.grad(sdf, xx, torch.ones_like(sdf), create_graph=True)
        return ((gx.norm(dim=-1) - 1) ** 2).mean()

    def run():
        xd = x.to(DEV).requires_grad_(True)
        l = eik(lambda fs, xx: ops.encode(xx, fs, meta), fd, xd, wd, bd)
        return l, torch.autograd.grad(l, fd + [xd])

    assert n >= ops.ENCODE_PULL_MIN_POINTS
    called = []
    real = ops._lib.load().miso_grad_pull_dx
    la, ga = run()
    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", None)
    lb, gb = run()
    assert abs(la.item() - lb.item()) <= 1e-6 * max(1.0, abs(lb.item()))
    for a, c in zip(ga, gb):
        assert relerr(a, c) < 5e-5
    # oracle on the first 300 points only (the any-order restatement is slow); gradient w.r.t. x
    # of those points depends on nothing else
    m = 300
    fc = [f.clone().requires_grad_(True) for f in feats]
    xc = x[:m].clone().requires_grad_(True)
    lc = eik(lambda fs, xx: R.encode_gather(fs, bound, xx), fc, xc, ws, bs)
    (gxc,) = torch.autograd.grad(lc, [xc])
    assert relerr(ga[-1][:m].cpu() * (n / m), gxc) < 5e-4


@pytest.mark.parametrize("name", ["small", "cfg2"])
@pytest.mark.parametrize("binned", [False, True])
def test_lerp_tree_encode_kernels_equal_weight_form(name, binned, monkeypatch):
    """encode_bwd_x_kernel / encode_bwd2_lean_kernel (value, first and mixed second derivatives of the interpolant by
    a lerp tree) against the weight-form kernels they replace when no grid gradient is scattered from the launch:
    grad_x of the first backward; gg_out and grad_x of the second one, with and without a cotangent of the grid
    gradient, points outside the bound and on its faces included (gridsample_cuda.cu:443-531)."""
    from miso_amd import ops
    case, feats, bound, ws, bs, x0, meta, fd, pack = setup_case(name)
    fdd = [f.detach() for f in fd]
    g = torch.Generator().manual_seed(5)
    b = torch.tensor(case["bound"])
    n = 20000
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.2 + b[:, 0] - 0.1 * (b[:, 1] - b[:, 0])
    x[:64] = b[:, 0]
    x[64:128] = b[:, 1]
    x = x.to(DEV)
    F = sum(f.shape[1] for f in fdd)
    gout = torch.randn(n, F, generator=g).to(DEV)
    ggx = torch.randn(n, 3, generator=g).to(DEV)
    ggf_all = [torch.randn(f.shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last_3d) * 1e-2
               for f in fdd]
    sb = ops.SortedBatch(n, x.device).sort(x, meta) if binned else None
    none = [False] * len(fdd)

    def run():
        gx1, _ = ops.encode_bwd_raw(x, fdd, meta, gout, True, none, sorted_batch=sb)
        outs = [gx1]
        for ggf in (None, ggf_all, [ggf_all[0]] + [None] * (len(fdd) - 1)):
            for e in (ggx, None):
                if e is None and ggf is None:
                    continue
                ggo, gx2, _ = ops.encode_bwd2_raw(x, fdd, meta, gout, e, ggf, True, none, sorted_batch=sb)
                outs += [ggo, gx2]
        torch.cuda.synchronize()
        return outs

    lean = run()
    monkeypatch.setenv("MISO_ENCODE_NO_LEAN", "1")
    ref = run()
    assert len(lean) == len(ref) == 11
    for a, c in zip(lean, ref):
        assert torch.isfinite(a).all()
        assert relerr(a, c) < 2e-5


def test_capi_rejects_bad_arguments_before_launching():
    """Return codes of the C ABI for malformed calls: nothing is launched,
    nothing crashes, the error string is meaningful."""
    import ctypes as C
    from miso_amd import _lib, ops
    lib = _lib.load()
    case, feats, bound, ws, bs, x0, meta, fd, pack = setup_case("small")
    fdd = [f.detach() for f in fd]
    x = x0.to(DEV)
    n = x.shape[0]
    g = ops._fill_grid(fdd, meta)
    out = torch.empty(n, sum(f.shape[1] for f in fdd), device=DEV)
    st = ops._stream(x)
    BAD, UNSUP = _lib.E_BADARG, _lib.E_UNSUPPORTED
    assert lib.miso_encode_fwd(C.byref(g), ops._ptr(x), -1, ops._ptr(out), out.stride(0), st) == BAD
    assert lib.miso_encode_fwd(C.byref(g), None, n, ops._ptr(out), out.stride(0), st) == BAD
    assert lib.miso_encode_fwd(C.byref(g), ops._ptr(x), n, ops._ptr(out), 1, st) == BAD          # ld < F
    g2 = ops._fill_grid(fdd, meta)
    g2.flags = 1 << 20                                                                           # unknown flag
    assert lib.miso_encode_fwd(C.byref(g2), ops._ptr(x), n, ops._ptr(out), out.stride(0), st) == BAD
    g3 = ops._fill_grid(fdd, meta)
    g3.n_levels = 0
    assert lib.miso_encode_fwd(C.byref(g3), ops._ptr(x), n, ops._ptr(out), out.stride(0), st) == BAD
    # sort: tiles out of range, misaligned normalised buffer
    sb = ops.SortedBatch(n, DEV)
    assert lib.miso_sort_points(C.byref(g), ops._ptr(x), n, 17, ops._ptr(sb.workspace), None, ops._ptr(sb.xn_sorted),
                                ops._ptr(sb.perm), ops._ptr(sb.tile_offsets), st) == BAD
    assert lib.miso_sort_points(C.byref(g), ops._ptr(x), n, 16, ops._ptr(sb.workspace), None,
                                C.c_void_p(sb.xn_sorted.data_ptr() + 4), ops._ptr(sb.perm),
                                ops._ptr(sb.tile_offsets), st) == BAD
    # fused: decoder shape outside the table is "unsupported", not a crash
    m, packed = pack.get()
    m2 = _lib.Mlp()
    C.memmove(C.byref(m2), C.byref(m), C.sizeof(m2))
    m2.hidden_dim = 48
    sdf = torch.empty(n, 1, device=DEV)
    assert lib.miso_sdf_fwd(C.byref(g), C.byref(m2), ops._ptr(packed), ops._ptr(x), n, ops._ptr(sdf), None, st) == UNSUP
    assert lib.miso_lm_normal_eq(ops._ptr(x), ops._ptr(x), ops._ptr(x), ops._ptr(sdf), ops._ptr(sdf), n, 7, 0.1,
                                 ops._ptr(sdf), st) == UNSUP
    assert b"argument" in lib.miso_error_string(BAD).lower() or len(lib.miso_error_string(BAD)) > 0
    # round 6 entries: the atlas query, the d-feat rows of the first backward, the pooling
    host = (C.c_char * int(lib.miso_atlas_plan_bytes(1)))()
    assert lib.miso_atlas_plan_bytes(0) == 0
    assert lib.miso_atlas_plan_build(C.byref(g), 0, C.cast(host, C.c_void_p)) == BAD
    assert lib.miso_atlas_plan_build(C.byref(g), 1, None) == BAD
    assert lib.miso_atlas_plan_build(C.byref(g), 1, C.cast(host, C.c_void_p)) == 0
    plan = torch.frombuffer(host, dtype=torch.uint8).clone().to(DEV)
    poses = torch.tensor([[1., 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]], device=DEV)
    args = lambda **kw: [kw.get("plan", ops._ptr(plan)), kw.get("S", 1), C.byref(g), kw.get("poses", ops._ptr(poses)),     # noqa: E731
                         kw.get("mlp", C.byref(m)), ops._ptr(packed), kw.get("x", ops._ptr(x)), kw.get("n", n), None, None,
                         None, kw.get("nx", 0), 0, 0, kw.get("sdf", ops._ptr(sdf)), None, 0, kw.get("flags", 0), st]
    assert lib.miso_atlas_sdf_fwd(*args(plan=None)) == BAD
    assert lib.miso_atlas_sdf_fwd(*args(poses=None)) == BAD
    assert lib.miso_atlas_sdf_fwd(*args(sdf=None)) == BAD                      # neither sdf nor feats asked for
    assert lib.miso_atlas_sdf_fwd(*args(flags=1)) == BAD                       # only MISO_F_EXACT_F32 is a flag here
    assert lib.miso_atlas_sdf_fwd(*args(x=None)) == BAD                        # no points and no lattice
    assert lib.miso_atlas_sdf_fwd(*args(mlp=C.byref(m2))) == UNSUP
    assert lib.miso_atlas_sdf_fwd(*args()) == 0
    rows = torch.empty(n, out.shape[1], device=DEV)
    mask = torch.zeros(((n + 63) // 64) * 64 * int(lib.miso_sdf_mask_words(C.byref(m))), device=DEV, dtype=torch.int32)
    gs = torch.ones(n, 1, device=DEV)
    assert lib.miso_sdf_bwd_rows(C.byref(g), C.byref(m), ops._ptr(packed), ops._ptr(x), n, None, ops._ptr(mask), None,
                                 ops._ptr(rows), st) == BAD
    assert lib.miso_sdf_bwd_rows(C.byref(g), C.byref(m), ops._ptr(packed), ops._ptr(x), n, ops._ptr(gs), ops._ptr(mask), None,
                                 C.c_void_p(rows.data_ptr() + 4), st) == BAD                       # rows must be 16-B aligned
    g4 = ops._fill_grid(fdd, meta)
    g4.flags = _lib.F_GRAD_OVERWRITE
    assert lib.miso_sdf_bwd_rows(C.byref(g4), C.byref(m), ops._ptr(packed), ops._ptr(x), n, ops._ptr(gs), ops._ptr(mask), None,
                                 ops._ptr(rows), st) == BAD                                        # binned-path flags do not apply
    bm = (C.c_float * 3)(0.0, 0.0, 0.0)
    pooled, cnt = torch.empty(8, 3, device=DEV), torch.empty(8, device=DEV, dtype=torch.int32)
    feats3 = torch.ones(n, 3, device=DEV)
    assert lib.miso_grid_pool_avg(ops._ptr(x), ops._ptr(feats3), n, 3, 3, bm, 0.0, 2, 2, 2, ops._ptr(pooled), ops._ptr(cnt), st) == BAD
    assert lib.miso_grid_pool_avg(ops._ptr(x), ops._ptr(feats3), n, 3, 2, bm, 0.5, 2, 2, 2, ops._ptr(pooled), ops._ptr(cnt), st) == BAD
    assert lib.miso_grid_pool_avg(ops._ptr(x), ops._ptr(feats3), n, 3, 3, bm, 0.5, 2, 2, 0, ops._ptr(pooled), ops._ptr(cnt), st) == BAD
    assert lib.miso_grid_pool_avg(ops._ptr(x), ops._ptr(feats3), n, 3, 3, bm, 0.5, 2, 2, 2, ops._ptr(pooled), ops._ptr(cnt), st) == 0
    torch.cuda.synchronize()
    assert int(cnt.sum()) == n and bool(((pooled == 1.0) | (pooled == 0.0)).all())
    # and a well-formed call still works afterwards
    assert lib.miso_encode_fwd(C.byref(g), ops._ptr(x), n, ops._ptr(out), out.stride(0), st) == 0
    torch.cuda.synchronize()


def test_full_size_training_step_vs_cpu_oracle_cfg2():
    """BASELINE cfg-2 at FULL size (262 144 points) against the CPU oracle directly: the binned
    training step (sort, forward + L1 loss, backward, pull) gives the reference's loss (stock ATen
    grid_sample per level, cat, nn.Linear chain, mean |sdf - target|) and its gradient for every
    level; the SDF agrees to 1e-5 everywhere (north-star tolerance)."""
    from miso_amd import ops
    from miso_amd.step import MappingStep
    case = gc.CASES["cfg2"]
    n = 262144
    gen = torch.Generator().manual_seed(77)
    x = torch.rand(n, 3, generator=gen) * 2 - 1
    target = torch.rand(n, 1, generator=gen) * 0.2 - 0.1
    bound = torch.tensor(case["bound"], dtype=torch.float32)
    shapes = [gc.grid_shape(case["bound"], c, case["fdim"]) for c in gc.level_cells(case)]
    feats = [torch.randn(s, generator=gen) * 1e-2 for s in shapes]
    sd = {k: T(v) for k, v in gc.make_decoder(case).items()}
    ws, bs = R.decoder_params(sd)
    # reference on the host
    fc = [f.clone().requires_grad_(True) for f in feats]
    pred = R.sdf_stock(fc, bound, x, ws, bs)
    loss = R.miso_loss_regression(pred, target, None, None, "L1")
    gref = torch.autograd.grad(loss, fc)
    # HIP path
    meta = ops.GridMeta.from_bound(bound)
    fd = [f.to(DEV).contiguous(memory_format=torch.channels_last_3d) for f in feats]
    pack = ops.DecoderPack([w.to(DEV) for w in ws], [b.to(DEV) for b in bs])
    step = MappingStep(fd, meta, pack, n, "L1", 1.0, 0.0, 0.0, use_graph=True)
    assert step.sorted is not None
    step.set_batch(x.to(DEV), target.to(DEV))
    step.run(); step.run()
    torch.cuda.synchronize()
    d = (step.sdf.cpu() - pred.detach()).abs()
    assert d.max().item() <= 1e-5 and d.mean().item() <= 1e-6
    assert abs(step.loss.sum().item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
    # Gradients in the Euclidean norm here; in the MAX norm -- with the census of the ReLU pre-activations that land
    # within fp32 rounding of zero, where two implementations may gate differently -- in
    # tests/test_config_shapes.py::test_full_size_cfg2_gradient_outliers_are_relu_ties
    for a, b in zip(step.grads, gref):
        a = a.cpu()
        assert ((a - b).double().norm() / b.double().norm()).item() < 5e-4
    # ... and the EXACT configuration bench.py times (the headline): stream launches, no SDF write-back, a batch sorted
    # without perm[] (the index rides in xn_sorted[:, 3]) -- against the same host evaluation
    head = MappingStep(fd, meta, pack, n, "L1", 1.0, 0.0, 0.0, sort=True, keep_sdf=False, use_graph=False)
    assert head._fused_train() and head.sorted is not None and head.sorted.perm is None
    head.set_batch(x.to(DEV), target.to(DEV))
    head.run(); head.run()
    torch.cuda.synchronize()
    assert abs(head.loss.sum().item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
    for a, b in zip(head.grads, gref):
        a = a.cpu()
        assert ((a - b).double().norm() / b.double().norm()).item() < 5e-4


@pytest.mark.parametrize("lt", ["L2", "L1"])
@pytest.mark.parametrize("level", [0, 1])
def test_pair_latent_kernel_vs_oracle_large(lt, level):
    """miso_pair_latent at an alignment-sized problem (200 000 source vertices, two ScanNet-shaped
    levels) against the oracle's pairwise_latent_loss + autograd: loss and the gradient w.r.t. all
    four pose tensors."""
    from miso_amd import ops
    g = torch.Generator().manual_seed(31 + level)
    bound = torch.tensor([[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]])
    shapes = [(1, 4, 20, 10, 20), (1, 4, 50, 25, 50)]       # (1,C,Z,Y,X): cells 1.0 / 0.4 m over 20x10x20 m
    fs = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    fdst = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    n = 200000
    coords = (torch.rand(n, 3, generator=g) - 0.5) * (bound[:, 1] - bound[:, 0])
    R_s = torch.tensor(gc.rodrigues(np.array([0.02, -0.03, 0.05])), dtype=torch.float32)
    R_d = torch.tensor(gc.rodrigues(np.array([-0.04, 0.01, 0.02])), dtype=torch.float32)
    t_s, t_d = torch.tensor([[0.3], [-0.2], [0.1]]), torch.tensor([[6.0], [0.4], [-1.5]])
    # oracle
    ps = [p.clone().requires_grad_(True) for p in (R_s, t_s, R_d, t_d)]
    ref = R.pairwise_latent_loss(fs, bound, fdst, bound, coords, *ps, level=level, fdim=4, align_weight=1.0,
                                 align_loss=lt)
    gref = torch.autograd.grad(ref, ps)
    # HIP
    nlv = level + 1
    meta = ops.GridMeta.from_bound(bound)
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last_3d)
    f_src = ops.encode(coords.to(DEV), [cl(f) for f in fs[:nlv]], meta).detach()
    pd = [p.detach().clone().to(DEV).requires_grad_(True) for p in (R_s, t_s, R_d, t_d)]
    val = ops.pair_latent(pd[0], pd[1], pd[2], pd[3], coords.to(DEV), f_src, [cl(f) for f in fdst[:nlv]], meta, lt)
    gg = torch.autograd.grad(val, pd)
    assert abs(val.item() - ref.item()) <= 2e-5 * abs(ref.item())
    for a, b in zip(gg, gref):
        assert relerr(a.cpu(), b) < 2e-3


@pytest.mark.parametrize("tiles", [16, (20, 16, 24)])
@pytest.mark.parametrize("n", [0, 1, 64, 65])
def test_binned_path_ragged_and_empty(n, tiles):
    """Binned entry points at the edges: an empty batch (every gradient must come back exactly zero
    -- MISO_F_GRAD_OVERWRITE promises no stale values), one point, one chunk, one chunk + 1.  Also under a per-axis
    binning, which the matrix-core pull alone serves (an empty batch used to come back as an error there)."""
    from miso_amd import ops
    case, feats, bound, ws, bs, x0, meta, fd, pack = setup_case("cfg2")
    fdd = [f.detach() for f in fd]
    L = len(fdd)
    x = x0[:n].to(DEV)
    sb = ops.SortedBatch(n, DEV, tiles=tiles).sort(x, meta)
    off = sb.tile_offsets.cpu()
    assert off[0] == 0 and off[-1] == n
    sdf, mask = ops.sdf_fwd_raw(x, fdd, meta, pack, True, sorted_batch=sb)
    assert sdf.shape == (n, 1)
    grads = [torch.full_like(f, 7.0) for f in fdd]               # poison: must be overwritten
    gs = torch.full((n, 1), 0.5, device=DEV)
    ops.sdf_bwd_raw(x, fdd, meta, pack, gs, mask, False, [True] * L, grads, sorted_batch=sb, overwrite=True)
    torch.cuda.synchronize()
    if n == 0:
        for gr in grads:
            assert float(gr.abs().max()) == 0.0
        return
    ref_s, ref_m = ops.sdf_fwd_raw(x, fdd, meta, pack, True)
    _, ref_g = ops.sdf_bwd_raw(x, fdd, meta, pack, gs, ref_m, False, [True] * L)
    assert torch.equal(sdf, ref_s)
    for a, b in zip(grads, ref_g):
        assert relerr(a, b) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,K", [(0, 1), (1, 1), (777, 5), (100000, 31)])
def test_rigid_by_index_vs_oracle(n, K):
    """miso_rigid_by_index against the reference's per-keyframe loop (oracle.transform_by_keyframe_loop), forward,
    transposed (the cotangent of the points), and through the autograd node the losses use (pose cotangents)."""
    from miso_amd import ops
    from miso_amd.grid_opt.loss import rigid_by_index
    from oracle import ref_torch as R_
    import golden_cases as gc_
    rs = np.random.RandomState(n + K)
    Rm = torch.from_numpy(np.stack([gc_.rodrigues(rs.uniform(-1, 1, 3)) for _ in range(K)]).astype(np.float32))
    tm = torch.from_numpy(rs.uniform(-5, 5, (K, 3, 1)).astype(np.float32))
    idx = torch.from_numpy(rs.randint(0, K, size=n).astype(np.int64))
    x = torch.from_numpy(rs.uniform(-10, 10, (n, 3)).astype(np.float32))
    want = R_.transform_by_keyframe_loop(x, idx, Rm, tm)
    got = ops.rigid_by_index(Rm.to(DEV), tm.reshape(K, 3).to(DEV), idx.to(DEV), x.to(DEV))
    assert got.shape == (n, 3)
    if n:
        assert (got.cpu() - want).abs().max().item() <= 4e-6           # |R x + t| <= 22: a couple of ulp
    gT = ops.rigid_by_index(Rm.to(DEV), None, idx.to(DEV), x.to(DEV), transpose=True)
    wantT = torch.einsum("nji,nj->ni", Rm[idx], x)
    if n:
        assert (gT.cpu() - wantT).abs().max().item() <= 4e-6
    if n == 0:
        return
    # autograd node: cotangents of R, t, x against the loop differentiated by torch on the CPU
    Rc, tc, xc = Rm.clone().requires_grad_(True), tm.clone().requires_grad_(True), x.clone().requires_grad_(True)
    w = torch.from_numpy(rs.standard_normal((n, 3)).astype(np.float32))
    (R_.transform_by_keyframe_loop(xc, idx, Rc, tc) * w).sum().backward()
    Rg, tg, xg = (Rm.to(DEV).requires_grad_(True), tm.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True))
    (rigid_by_index(Rg, tg, idx.to(DEV), xg) * w.to(DEV)).sum().backward()
    for a, b in ((Rg.grad, Rc.grad), (tg.grad, tc.grad), (xg.grad, xc.grad)):
        assert (a.cpu() - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())


@pytest.mark.gpu
def test_adam_nan_guard_on_the_device():
    """DenseAdam.step(guard=loss): a NaN loss leaves parameters, moments, flags and (one call later) the step counts
    alone -- the reference's "Loss is nan! Skip backward step" (trainer.py:213-219) without a read-back -- and the
    steps around it are bit-identical to an optimizer that never saw the NaN step."""
    from miso_amd.optim import DenseAdam
    torch.manual_seed(0)
    big = torch.randn(3, 5000, device=DEV)
    tiny = torch.randn(1, 3, device=DEV)
    grads = [(torch.randn_like(big), torch.randn_like(tiny)) for _ in range(4)]

    def run(with_nan):
        pb, pt = big.clone().requires_grad_(True), tiny.clone().requires_grad_(True)
        opt = DenseAdam([pb, pt], lr=1e-2)
        for k, (gb, gt) in enumerate(grads):
            if with_nan and k == 2:                       # an extra, poisoned step in the middle
                pb.grad, pt.grad = torch.full_like(pb, float("nan")), torch.full_like(pt, float("nan"))
                opt.step(clear_grads=True, guard=torch.tensor(float("nan"), device=DEV))
                assert float(pb.grad.abs().sum()) == 0.0 and float(pt.grad.abs().sum()) == 0.0   # consumed all the same
            pb.grad, pt.grad = gb.clone(), gt.clone()
            opt.step(guard=torch.tensor(0.5, device=DEV))
        sd = opt.state_dict()
        return pb.detach(), pt.detach(), opt, sd

    a_b, a_t, opt_a, sd_a = run(False)
    b_b, b_t, opt_b, sd_b = run(True)
    assert torch.equal(a_b, b_b) and torch.equal(a_t, b_t)
    assert opt_a.skipped_steps == 0 and opt_b.skipped_steps == 1
    for sa, sb in zip(sd_a["state"].values(), sd_b["state"].values()):
        assert sa["step"] == sb["step"] == 4
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    assert torch.isfinite(b_b).all()


@pytest.mark.gpu
@pytest.mark.parametrize("loss_type", ["L2", "L1"])
def test_align_plan_iteration_vs_autograd(loss_type):
    """miso_align_iteration_a / _b (every pair in one launch, so3_exp_map backward in the epilogue, Adam on the
    device) against autograd: the per-pair fused op through miso_amd.so3.so3_exp_map for the pose gradients of ALL
    submaps, torch.optim.Adam for the step."""
    from test_grid_opt_mirror import make_atlas
    from miso_amd import ops
    import miso_amd.grid_opt.align.miso as AM
    from miso_amd.grid_opt.align.base import grid_atlas_pose_trust_region_loss
    dev = "cuda:0"
    atlas = make_atlas(dev)
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    S = atlas.num_submaps
    pairs = [(0, 1), (0, 2), (1, 2)]
    level = 1
    inputs = AM.latent_pair_inputs(atlas, pairs, level=level, fdim=4, check_intersection=True)
    R0 = torch.stack(list(atlas.R_world_submap_list))
    t0 = torch.stack(list(atlas.t_world_submap_list))
    plan = ops.AlignPlan(R0, t0, inputs, loss_type=loss_type, align_weight=3000.0, lr=1e-2, reg_weight=2.0,
                         reg_thresh_rad=1e-3, reg_thresh_m=1e-3, ring_iters=4, save_poses=True)
    dr = torch.cat([p.detach().reshape(1, 3) for p in atlas.rotation_corrections])
    dt = torch.cat([p.detach().reshape(1, 3) for p in atlas.translation_corrections])
    plan.params.copy_(torch.cat((dr, dt), 1))
    plan.iteration_a()
    # autograd reference
    atlas.zero_grad(set_to_none=True)
    losses = []
    for a, b in pairs:
        assert bool(atlas.check_submap_intersection(a, b))
        (v,) = AM.pairwise_loss_latent(atlas, None, a, b, level=level, fdim=4, align_loss=loss_type, device=dev).values()
        losses.append(v)
    total = sum(losses)
    total.backward()
    flat = plan.flat.cpu()
    torch.testing.assert_close(plan.pair_losses.cpu(), torch.stack(losses).detach().cpu(), rtol=2e-5, atol=1e-4)
    assert abs(flat[6 * S].item() - total.item()) <= 2e-5 * abs(total.item())
    g_ref = torch.cat((torch.cat([p.grad.reshape(1, 3) for p in atlas.rotation_corrections]),
                       torch.cat([p.grad.reshape(1, 3) for p in atlas.translation_corrections])), 1).cpu()
    scale = g_ref.abs().max().item()
    assert (flat[:6 * S].view(S, 6) - g_ref).abs().max().item() <= 2e-4 * scale, (flat[:6 * S].view(S, 6), g_ref)
    # poses written by the prologue = GridAtlas.updated_submap_pose
    for s in range(S):
        R, t = atlas.updated_submap_pose(s)
        torch.testing.assert_close(plan.poses[s, :9].view(3, 3), R.detach(), rtol=0, atol=2e-6)
        torch.testing.assert_close(plan.poses[s, 9:].view(3, 1), t.detach(), rtol=0, atol=1e-6)
    # second half: regulariser + Adam, against torch.optim.Adam on submaps 1..S-1
    params = [p for s in range(1, S) for p in atlas.params_for_submap_pose(s)]
    opt = torch.optim.Adam(params, lr=1e-2)
    reg = sum(grid_atlas_pose_trust_region_loss(atlas, thresh_rad=1e-3, thresh_m=1e-3, weight=2.0).values())
    reg.backward()
    opt.step()
    plan.iteration_b()
    prm = plan.params.cpu()
    for s in range(1, S):
        torch.testing.assert_close(prm[s, :3].reshape(1, 3), atlas.rotation_corrections[s].detach().cpu(), rtol=0, atol=2e-6)
        torch.testing.assert_close(prm[s, 3:].reshape(3, 1), atlas.translation_corrections[s].detach().cpu(), rtol=0, atol=2e-6)
    torch.testing.assert_close(prm[0], torch.cat((dr[0], dt[0])).cpu(), rtol=0, atol=0)       # submap 0 is fixed
    c = plan.ctrl()
    assert c == dict(steps=1, stopped=False, iterations=1, skipped=0)
    row = plan.ring()[0].cpu()
    assert abs(row[0].item() - (total + reg).item()) <= 2e-5 * abs(total.item()) and row[1].item() == float("inf")
    assert row[2:].view(S, 4, 4)[:, 3].tolist() == [[0.0, 0.0, 0.0, 1.0]] * S


@pytest.mark.gpu
def test_align_plan_rejects_bad_arguments():
    import ctypes as C
    from miso_amd import _lib
    lib = _lib.load()
    cfg = _lib.Align()
    cfg.n_submaps, cfg.n_pairs, cfg.loss_type = 0, 0, 2
    assert lib.miso_align_iteration_a(C.byref(cfg), None) == _lib.E_BADARG
    cfg.n_submaps = 65
    assert lib.miso_align_plan_build(None, C.byref(cfg), None) == _lib.E_BADARG
    assert lib.miso_align_state_layout(65, 1, 0, 0, None) == 0
    assert lib.miso_align_plan_bytes(-1) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["unsorted", "sorted_thin", "sorted_crowd"])
def test_adam_touched_flags_equal_gradient_scan(path):
    """miso_adam_touched (Adam driven by the flags the scatter kernels leave in miso_level_t.grad_touched) against
    miso_adam_active (Adam that reads the gradient to find what moved): parameters, both moments, the active flags and
    the cleared gradients bit for bit over four steps with different batches, one of them with a NaN loss guard; the
    flags are all cleared afterwards.  unsorted: float atomics from the decoder backward, gradients cleared by Adam.
    sorted_thin: owner-computes pull (overwrite).  sorted_crowd: matrix-core push + atomics of a fine level."""
    from miso_amd import ops
    torch.manual_seed(3)
    C, H = 4, 64
    sizes = [(40, 25, 40), (130, 70, 130)]
    meta = ops.GridMeta.from_bound([[-2.0, 2.0], [-1.0, 1.5], [-2.0, 2.0]])
    lin = [torch.nn.Linear(2 * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(DEV) for l in lin], [l.bias.data.to(DEV) for l in lin])
    n = {"unsorted": 3000, "sorted_thin": 70000, "sorted_crowd": 100 * 4096 + 5}[path]
    f0 = [(torch.randn(1, C, z, y, x, device=DEV) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
          for (x, y, z) in sizes]
    # two Adam states fed the SAME gradients (float atomics do not sum in a reproducible order): A scans them,
    # B reads the flags
    featsA, featsB = [f.clone() for f in f0], [f.clone() for f in f0]
    mA, vA = [torch.zeros_like(f) for f in f0], [torch.zeros_like(f) for f in f0]
    mB, vB = [torch.zeros_like(f) for f in f0], [torch.zeros_like(f) for f in f0]
    actA, actB = [ops.adam_active_flags(f) for f in f0], [ops.adam_active_flags(f) for f in f0]
    tch = [ops.adam_active_flags(f) for f in f0]
    grads = [torch.zeros_like(f) for f in f0]
    gen = torch.Generator().manual_seed(11)
    zero = path == "unsorted"
    for t in range(4):
        # batches wander: chunks wake up at different steps
        ctr = torch.tensor([-1.0 + 0.5 * t, 0.1 * t, 0.3 * t - 0.5])
        x = ((torch.rand(n, 3, generator=gen) - 0.5) * torch.tensor([1.2, 0.8, 1.0]) + ctr).to(DEV)
        gs = (torch.randn(n, 1, generator=gen) / n).to(DEV)
        guard = torch.tensor([float("nan") if t == 2 else 1.0], device=DEV)
        if path == "unsorted":
            _, mask = ops.sdf_fwd_raw(x, featsB, meta, pack, True)
            ops.sdf_bwd_raw(x, featsB, meta, pack, gs, mask, False, [True, True], grads, touched=tch)
        else:
            sb = ops.SortedBatch(n, DEV).sort(x, meta)
            _, mask = ops.sdf_fwd_raw(x, featsB, meta, pack, True, sorted_batch=sb)
            ops.sdf_bwd_raw(x, featsB, meta, pack, gs, mask, False, [True, True], grads, sorted_batch=sb,
                            overwrite=True, touched=tch)
        for g, tc in zip(grads, tch):            # every non-zero gradient lies in a flagged chunk
            nz = (torch.as_strided(g, (g.numel(),), (1,)) != 0).nonzero().reshape(-1)      # storage order
            assert bool(tc[nz // ops._lib.ADAM_CHUNK].all()) and int(tc.sum()) > 0
        gA = [g.clone() for g in grads]
        for i in range(2):
            ops.adam_active_(featsA[i], gA[i], mA[i], vA[i], actA[i], t + 1, 1e-3, zero_grad=zero, guard=guard)
            ops.adam_active_(featsB[i], grads[i], mB[i], vB[i], actB[i], t + 1, 1e-3, zero_grad=zero, guard=guard,
                             touched=tch[i])
        assert all(int(tc.sum()) == 0 for tc in tch)
        for a, b in zip(featsA + mA + vA + actA + gA, featsB + mB + vB + actB + grads):
            assert torch.equal(a, b)
        if zero:
            assert all(bool((g == 0).all()) for g in grads)
    assert 0 < float(actB[-1].float().mean()) < 0.9      # some chunks never woke up


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["small", "scannet"])
def test_lattice_overlap_gate_counts_equal_the_point_list_gate(case):
    """The overlap gate of the fused alignment iteration in its lattice form (a lane per lattice row solves for the
    in-bound index interval and evaluates the reference's per-vertex arithmetic only around the interval ends) must
    count exactly what the point-list form counts by evaluating every vertex: identity and axis-swapping rotations
    (rows exactly parallel to faces), translations that put vertices ON faces, small and large rotations, no overlap
    at all, containment, a NaN pose.  scannet: the real lattice (200 x 100 x 200 vertices of a 20 x 10 x 20 m bound,
    0.1 m cells) under poses like those of neighbouring submaps."""
    from miso_amd import ops
    from miso_amd.so3 import so3_exp_map
    torch.manual_seed(5)
    if case == "small":
        nx, ny, nz = 40, 18, 30
        bound = [[-2.0, 2.0], [-0.9, 0.9], [-1.5, 1.5]]
        rots = [torch.zeros(3), torch.tensor([0.0, 0.0, math.pi / 2]), torch.tensor([math.pi / 2, 0.0, 0.0]),
                torch.tensor([0.0, 1e-4, 0.0]), torch.tensor([0.02, -0.01, 0.03]), torch.tensor([0.4, 0.2, -0.7]),
                torch.tensor([1.2, -2.0, 0.6]), torch.tensor([0.0, math.pi, 0.0]), torch.randn(3), torch.zeros(3),
                # faces met at a fraction of a degree: many vertices of a row lie within the gate's slack of the face
                torch.tensor([0.0, 0.004, 0.0]), torch.tensor([0.002, -0.0007, 0.005])]
        trans = [torch.zeros(3), torch.tensor([0.05, 0.0, 0.0]), torch.tensor([3.95, 0.0, 0.0]), torch.tensor([0.0, 1.75, 0.0]),
                 torch.tensor([10.0, 0.0, 0.0]), torch.tensor([0.3, -0.2, 0.1]), torch.tensor([1.0, 0.9, -1.5]),
                 torch.tensor([-2.0, 0.0, 2.95]), torch.randn(3), torch.zeros(3),      # (coincides with the first)
                 torch.tensor([0.4, 0.1, -0.2]), torch.tensor([-1.0, 0.45, 0.7])]
    else:
        nx, ny, nz = 200, 100, 200
        bound = [[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]]
        rots = [torch.zeros(3), torch.tensor([0.0, 0.3, 0.0]), torch.tensor([0.01, -0.5, 0.02]), torch.tensor([0.0, 1e-5, 0.0]),
                torch.tensor([0.0, math.pi / 2, 0.0]), torch.randn(3) * 0.2, torch.randn(3) * 0.05, torch.zeros(3),
                torch.tensor([0.0, 0.004, 0.0]), torch.tensor([0.0015, -0.0004, 0.006])]      # 0.2 - 0.35 degrees
        trans = [torch.zeros(3), torch.tensor([6.0, 0.2, -3.0]), torch.tensor([-8.0, 0.0, 7.5]), torch.tensor([0.1, 0.0, 0.0]),
                 torch.tensor([19.9, 0.0, 0.0]), torch.randn(3) * 4, torch.randn(3) * 4, torch.tensor([35.0, 0.0, 0.0]),
                 torch.tensor([9.0, -0.3, 2.0]), torch.tensor([-4.0, 4.6, -11.0])]
    half = [(b[1] - b[0]) / (2 * n) for b, n in zip(bound, (nx, ny, nz))]
    axes = [torch.linspace(b[0] + h, b[1] - h, n) for b, h, n in zip(bound, half, (nx, ny, nz))]
    zz, yy, xx = torch.meshgrid(axes[2], axes[1], axes[0], indexing="ij")          # z-major, x fastest
    pts = torch.stack([xx, yy, zz], dim=-1).reshape(-1, 3).to(DEV)
    S = len(rots)
    R0 = so3_exp_map(torch.stack(rots)).to(DEV)
    t0 = torch.stack(trans).reshape(S, 3, 1).to(DEV)
    C_ = 4
    feat = (torch.randn(1, C_, 6, 5, 8) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    meta = ops.GridMeta.from_bound(bound)
    coords = pts[:64].contiguous()
    fsrc = torch.randn(64, C_, device=DEV)
    pairs = [(a, b) for a in range(S) for b in range(S) if a != b]
    counts = {}
    for lattice in (False, True):
        descr = [dict(src=a, dst=b, coords=coords, feats_src=fsrc, feats_dst=[feat], meta_dst=meta, gate_pts=pts,
                      **({"gate_dims": (nx, ny, nz)} if lattice else {})) for a, b in pairs]
        plan = ops.AlignPlan(R0, t0, descr, loss_type="L2")
        plan.iteration_a()
        counts[lattice] = plan.overlap_counts.clone().cpu()
        if lattice:      # a NaN pose: both forms must agree there too (nothing in bound)
            plan.params[3, 0] = float("nan")
            plan.iteration_a()
            nan_lat = plan.overlap_counts.clone().cpu()
        else:
            plan.params[3, 0] = float("nan")
            plan.iteration_a()
            nan_pts = plan.overlap_counts.clone().cpu()
    assert torch.equal(counts[True], counts[False]), (counts[True] - counts[False]).abs().max()
    assert torch.equal(nan_lat, nan_pts)
    c = counts[True]
    assert (c == 0).any() and ((c > 0) & (c < nx * ny * nz)).sum() > 20
    if case == "small":
        assert (c == nx * ny * nz).any()
    # and the reference's own formulation, on the host in fp32 (transform_points_to -> transfrom_points_from ->
    # coords_in_bound, grid_atlas.py:405-420), for a few pairs
    P = pts.cpu()
    b = torch.tensor(bound)
    for idx in (0, 7, 19, 33, 52, 80) if case == "small" else (0, 9, 23, 41):
        a_, b_ = pairs[idx]
        w = P @ R0[a_].cpu().T + t0[a_].cpu().reshape(1, 3)
        q = (w - t0[b_].cpu().reshape(1, 3)) @ R0[b_].cpu()
        ref = int(((q >= b[:, 0]) & (q <= b[:, 1])).all(dim=1).sum())
        assert abs(int(c[idx]) - ref) <= max(2, int(2e-4 * ref)), (idx, int(c[idx]), ref)     # matmul order differs


@pytest.mark.gpu
def test_box_culled_pair_stage_equals_reading_every_vertex():
    """miso_align_src_boxes + miso_align_pair_t.src_boxes: the residual kernel skips a run of 64 source vertices when its
    box, mapped into the destination frame, cannot touch the destination bound.  The skip is conservative, so all 24
    sums of every pair are those of the kernel that reads every vertex (same lanes add the same terms; only the fp64
    atomics' order is free): identity, faces touching, rotated overlaps, no overlap at all, far-away world origins (the
    slack has to cover the rounding of w = Rs p + ts), a NaN pose.  Both lists are checked: a lattice in z-major order
    (what precompute_coordinates_for_alignment leaves) and a shuffled one (boxes of scattered vertices: nothing may be
    skipped wrongly, little is skipped at all)."""
    from miso_amd import ops, _lib
    from miso_amd.so3 import so3_exp_map
    torch.manual_seed(7)
    nx, ny, nz = 50, 24, 40
    bound = [[-2.5, 2.5], [-1.2, 1.2], [-2.0, 2.0]]
    half = [(b[1] - b[0]) / (2 * n) for b, n in zip(bound, (nx, ny, nz))]
    axes = [torch.linspace(b[0] + h, b[1] - h, n) for b, h, n in zip(bound, half, (nx, ny, nz))]
    zz, yy, xx = torch.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
    pts = torch.stack([xx, yy, zz], dim=-1).reshape(-1, 3)
    pts = pts[torch.rand(pts.shape[0]) > 0.3].contiguous()           # compacted, as the norm threshold leaves it
    rots = [torch.zeros(3), torch.tensor([0.0, 0.0, math.pi / 2]), torch.tensor([0.02, -0.01, 0.03]),
            torch.tensor([0.4, 0.2, -0.7]), torch.tensor([1.2, -2.0, 0.6]), torch.randn(3), torch.zeros(3), torch.zeros(3)]
    trans = [torch.zeros(3), torch.tensor([4.98, 0.0, 0.0]), torch.tensor([0.0, 2.39, 0.0]), torch.tensor([0.3, -0.2, 0.1]),
             torch.tensor([1.0, 0.9, -1.5]), torch.randn(3), torch.tensor([40.0, 0.0, 0.0]), torch.tensor([2.0, 0.5, 3.9])]
    S = len(rots)
    C_ = 4
    feats = [(torch.randn(1, C_, 8, 6, 10) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d),
             (torch.randn(1, C_, 16, 12, 20) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)]
    meta = ops.GridMeta.from_bound(bound)
    pairs = [(a, b) for a in range(S) for b in range(S) if a != b]
    for order, far in (("zmajor", 0.0), ("shuffled", 0.0), ("zmajor", 3000.0)):
        coords = (pts if order == "zmajor" else pts[torch.randperm(pts.shape[0])]).to(DEV).contiguous()
        n = coords.shape[0]
        fsrc = torch.randn(n, 2 * C_, device=DEV)
        # the box table itself
        bx = torch.empty(((n + 63) // 64, 6), device=DEV)
        assert _lib.load().miso_align_src_boxes(ops._ptr(coords), n, ops._ptr(bx), ops._stream(coords)) == 0
        pad = torch.cat([coords, coords[-1:].expand(bx.shape[0] * 64 - n, 3)]).view(-1, 64, 3)
        assert torch.equal(bx, torch.cat([pad.amin(1), pad.amax(1)], dim=1))
        R0 = so3_exp_map(torch.stack(rots)).to(DEV)
        t0 = (torch.stack(trans) + far).reshape(S, 3, 1).to(DEV)      # far: every submap 3 km from the world origin
        out = {}
        for cull in (False, True):
            descr = [dict(src=a, dst=b, coords=coords, feats_src=fsrc, feats_dst=feats, meta_dst=meta, gate_pts=None)
                     for a, b in pairs]
            plan = ops.AlignPlan(R0, t0, descr, loss_type="L2", cull=cull)
            plan.iteration_a()
            o = [plan.pair_out.clone().cpu()]
            plan.params[2, 0] = 0.3
            plan.params[4, 3:] = torch.tensor([0.2, -0.1, 0.05], device=DEV)
            plan.iteration_a()
            o.append(plan.pair_out.clone().cpu())
            plan.params[3, 1] = float("nan")
            plan.iteration_a()
            o.append(plan.pair_out.clone().cpu())
            out[cull] = o
        for a_, b_ in zip(out[False], out[True]):
            assert torch.equal(torch.isnan(a_), torch.isnan(b_))
            assert torch.equal(a_[:, 1], b_[:, 1])                      # in-bound counts: exact
            torch.testing.assert_close(torch.nan_to_num(b_), torch.nan_to_num(a_), rtol=1e-12, atol=1e-12)
        cnt = out[True][0][:, 1]
        assert (cnt == 0).any() and (cnt > 0).any()
    # (that runs ARE skipped shows in the time of the cfg-4 level-1 iteration: bench.py)


@pytest.mark.gpu
def test_mapping_batch_matches_the_op_sequence_and_sanitises():
    """miso_mapping_batch = table lookup + rigid_by_index + the interleaved label rows, with strided / bool columns; with
    sanitize it equals the same on torch.nan_to_num'ed inputs (what prepare_batch does, NaN -> 0, inf -> +-FLT_MAX)."""
    from miso_amd import ops
    g = torch.Generator().manual_seed(8)
    n, K = 5000, 7
    from miso_amd.so3 import so3_exp_map
    R = so3_exp_map(torch.randn(K, 3, generator=g)).to(DEV).contiguous()
    t = torch.randn(K, 3, generator=g).to(DEV)
    table = torch.tensor([3, -1, 0, 6, 2, 5, 1, 4, -1], dtype=torch.int64, device=DEV)          # key -> pose index
    fid = torch.randint(-2, 11, (n, 1), generator=g).to(DEV)                                    # some out of the table
    x = torch.randn(n, 3, generator=g).to(DEV)
    block = torch.randn(n, 4, generator=g).to(DEV)                                              # strided columns
    valid = (torch.rand(n, 1, generator=g) > 0.2).to(DEV)
    for sanitize in (False, True):
        xs, bl = x.clone(), block.clone()
        if sanitize:
            xs[5, 1] = float("nan"); xs[9, 0] = float("inf"); bl[7, 0] = float("nan"); bl[11, 3] = float("-inf")
        xo, ro = torch.empty(n, 3, device=DEV), torch.empty(n, 4, device=DEV)
        ops.mapping_batch(R, t, table, fid, xs, bl[:, 0:1], valid, bl[:, 2:3], bl[:, 3:4], xo, ro, sanitize=sanitize)
        xc, bc = (torch.nan_to_num(xs), torch.nan_to_num(bl)) if sanitize else (xs, bl)
        idx = table[fid[:, 0].clamp(0, table.numel() - 1)].clamp(0, K - 1)
        want_x = ops.rigid_by_index(R, t, idx, xc)
        assert torch.equal(xo, want_x)
        want_rows = torch.cat([bc[:, 0:1], valid.float(), bc[:, 2:3], bc[:, 3:4]], dim=1)
        assert torch.equal(ro, want_rows)
        assert torch.isfinite(xo).all() == sanitize or not sanitize


@pytest.mark.gpu
def test_round2_entry_points_on_empty_and_tiny_batches():
    """Edges of the entry points added for the captured trainer step and the tracker: an empty batch launches nothing
    that reads a row (loss slots cleared, step count untouched by a NaN-free empty loss), a 1-row and a 65-row batch
    (one lane of the second wavefront) equal the op-by-op path."""
    from miso_amd import ops
    C_, H = 4, 64
    feats = [(torch.randn(1, C_, 6, 5, 8) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d),
             (torch.randn(1, C_, 12, 10, 16) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)]
    meta = ops.GridMeta.from_bound([[-1.0, 1.0], [-0.5, 0.5], [-0.8, 0.8]])
    lin = [torch.nn.Linear(2 * C_, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(DEV) for l in lin], [l.bias.data.to(DEV) for l in lin])
    g = torch.Generator().manual_seed(1)
    for n in (0, 1, 65):
        x = (torch.rand(n, 3, generator=g) * 2 - 1).to(DEV) * torch.tensor([1.0, 0.5, 0.8], device=DEV)
        aux = torch.stack([torch.randn(n, generator=g) * 0.1, torch.ones(n), (torch.rand(n, generator=g) > 0.5).float(),
                           torch.rand(n, generator=g) + 0.5], dim=1).to(DEV).contiguous()
        mw = ops.sdf_mask_words(pack)
        mask = torch.zeros(((n + 63) // 64) * 64 * mw + 1, device=DEV, dtype=torch.int32)
        gsdf = torch.full((max(n, 1),), 7.0, device=DEV)[:n]
        slots = torch.full((ops._lib.LOSS_SLOTS, 2), 5.0, device=DEV)
        sdf = torch.empty(n, 1, device=DEV)
        ops.sdf_fwd_loss_unsorted_raw(x.contiguous(), feats, meta, pack, aux, mask, gsdf, slots, "L1", 1.0, 0.3, 0.2,
                                      sdf_out=sdf)
        if n == 0:
            assert float(slots.abs().sum()) == 0.0
            continue
        want_sdf, _ = ops.sdf_fwd_raw(x.contiguous(), feats, meta, pack, True)
        assert torch.allclose(sdf, want_sdf, rtol=1e-6, atol=1e-7)
        terms, gref = ops.mapping_loss_raw(want_sdf, aux[:, 0:1].contiguous(), aux[:, 1:2].contiguous(),
                                           aux[:, 2:3].contiguous(), aux[:, 3:4].contiguous(), "L1", 1.0, 0.3, 0.2)
        assert torch.allclose(slots.sum(0), terms, rtol=1e-5, atol=1e-8)
        assert torch.allclose(gsdf, gref.reshape(-1), rtol=1e-6, atol=1e-9)
    # mapping_batch with no rows: nothing written, no error
    R = torch.eye(3, device=DEV).reshape(1, 3, 3).contiguous()
    t = torch.zeros(1, 3, device=DEV)
    table = torch.zeros(2, dtype=torch.int64, device=DEV)
    e3, e4 = torch.empty(0, 3, device=DEV), torch.empty(0, 4, device=DEV)
    ops.mapping_batch(R, t, table, torch.empty(0, 1, dtype=torch.int64, device=DEV), e3, torch.empty(0, 1, device=DEV),
                      None, None, None, e3.clone(), e4, sanitize=True)


@pytest.mark.gpu
def test_adam_device_step_host_ring_and_multi_tensor_launch():
    """ops.AdamDeviceStep: the loss total + step count launch (a NaN total leaves the count alone; the total reaches the
    host through the pinned ring, each slot tagged with the number of the launch that wrote it -- what the trainer's NaN
    guard polls instead of copying), and several tensors stepped by ONE launch equal one launch per tensor bit for bit."""
    from miso_amd import ops
    # loss total + step count: a NaN total leaves the count alone
    dev = ops.AdamDeviceStep(1e-3, 0.9, 0.999, 1e-8, DEV, count=3)
    tot = torch.zeros((), device=DEV)
    host0 = dev.note_launch()
    assert not host0.ready()
    dev.total_and_bump(torch.full((ops._lib.LOSS_SLOTS, 2), 0.25, device=DEV), tot)
    assert float(tot) == 0.25 * 2 * ops._lib.LOSS_SLOTS and dev.step.tolist() == [4, 1]
    bad = torch.full((ops._lib.LOSS_SLOTS, 2), 0.25, device=DEV)
    bad[3, 1] = float("nan")
    host1 = dev.note_launch()
    dev.total_and_bump(bad, tot)
    # the count stays, the launch counter moves; the pinned ring (written by the kernel) holds both totals, each with the
    # number of the launch that wrote it
    assert bool(torch.isnan(tot)) and dev.step.tolist() == [4, 2]
    assert host0.ready() and host0.value() == 0.25 * 2 * ops._lib.LOSS_SLOTS
    assert host1.ready() and math.isnan(host1.value()) and host1.seq == 2
    assert host0.slot == 0 and host1.slot == 1
    for _ in range(dev.RING - 1):                    # the ring wraps to the slot it started from
        last = dev.note_launch()
        dev.total_and_bump(torch.full((ops._lib.LOSS_SLOTS, 2), 0.5, device=DEV), tot)
    torch.cuda.synchronize()
    assert last.slot == host0.slot and last.ready() and not host0.ready()
    assert last.value() == 0.5 * 2 * ops._lib.LOSS_SLOTS
    dev.set_count(7)
    assert dev.step.tolist() == [7, dev.RING + 1]
    assert dev.table.shape[1] == 6 and dev.rows > 20000
    # several tensors in one launch == one launch per tensor, bit for bit (ragged sizes, a gradient cleared or kept, chunks
    # that never moved, a NaN guard that skips the step)
    g = torch.Generator().manual_seed(3)
    sizes = [70000, 4096 + 37, 64, 1 << 20]
    def fresh():
        ts = []
        for i, nfl in enumerate(sizes):
            p = torch.randn(nfl, generator=g).to(DEV)
            gr = torch.randn(nfl, generator=g).to(DEV)
            gr[nfl // 3: 2 * nfl // 3] = 0.0                                   # chunks without a gradient
            m, v = torch.zeros(nfl, device=DEV), torch.zeros(nfl, device=DEV)
            tch = None
            if i == 3:      # the big one is stepped by `touched` flags, as the scatter kernels would have left them
                tch = ops.adam_active_flags(p)
                tch.copy_((gr.reshape(-1, ops._lib.ADAM_CHUNK) != 0).any(dim=1).to(torch.uint8))
            ts.append([p, gr, m, v, ops.adam_active_flags(p), i % 2 == 0, tch])
        return ts
    g.manual_seed(3)
    one = fresh()
    g.manual_seed(3)
    many = fresh()
    devs = [ops.AdamDeviceStep(1e-3, 0.9, 0.999, 1e-8, DEV, count=0) for _ in range(2)]
    packed = devs[1].multi([tuple(t) for t in many])
    for it, guard_val in enumerate((0.5, 0.25, float("nan"), 0.125)):
        guard = torch.full((1,), guard_val, device=DEV)
        for d_ in devs:
            d_.bump(guard)
        for p, gr, m, v, act, zero, tch in one:
            devs[0].step_(p, gr, m, v, act, zero_grad=zero, guard=guard, touched=tch)
        devs[1].step_multi_(packed, guard=guard)
        for a_, b_ in zip(one, many):
            for x_, y_ in zip(a_[:5], b_[:5]):
                assert torch.equal(x_, y_)
            if a_[6] is not None:
                assert torch.equal(a_[6], b_[6]) and int(a_[6].sum()) == 0      # consumed: the flags are cleared
            if not a_[5]:      # a gradient that was kept: give both the next step's
                new = torch.randn(a_[1].shape, generator=g).to(DEV)
                a_[1].copy_(new), b_[1].copy_(new)
            else:
                assert int((a_[1] != 0).sum()) == 0
                new = torch.randn(a_[1].shape, generator=g).to(DEV)
                a_[1].copy_(new), b_[1].copy_(new)
            if a_[6] is not None:      # the next step's flags for the new gradient
                fl = (a_[1].reshape(-1, ops._lib.ADAM_CHUNK) != 0).any(dim=1).to(torch.uint8)
                a_[6].copy_(fl), b_[6].copy_(fl)
    assert devs[0].step.tolist()[0] == 3 and float(one[0][2].abs().sum()) > 0
