"""Randomised cross-check of the binned path (sort -> tile-ordered gathers -> owner-computes
pull) against the plain atomic path on shapes the fixed cases do not reach: non-cubic grids,
sizes that 16 tiles do not divide, tiny grids (bricks of 0 or 1 vertices), mixed C, clustered and
out-of-bound points, ignored levels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("mc", [True, False], ids=["matrix_core", "vector"])
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MISO_STRESS_SEEDS", "14"))))
def test_binned_encode_matches_atomic(seed, mc, monkeypatch):
    """mc: the matrix-core pull (grad_pull_mc.hip: taken for levels at least as fine as the binning) or, with
    MISO_PULL_MC=0, the vector kernels of grad_pull.hip for every level (they stay the path of coarser grids and of the
    second-order gradient)."""
    from miso_amd import ops
    if not mc:
        monkeypatch.setenv("MISO_PULL_MC", "0")
    rs = np.random.RandomState(1000 + seed)
    L = int(rs.randint(1, 5))
    C = int(rs.choice([4, 8]))
    dims = [tuple(int(v) for v in rs.choice([3, 8, 12, 16, 20, 33, 48, 64, 100, 128], size=3)) for _ in range(L)]
    if seed % 5 == 0:
        dims = [(16 * (l + 1),) * 3 for l in range(L)]          # the divisible fast path, cubic
    bmin = rs.uniform(-3, 0, size=3)
    bmax = bmin + rs.uniform(0.5, 6, size=3)
    bound = [[float(bmin[a]), float(bmax[a])] for a in range(3)]
    n = int(rs.choice([16384, 20011, 50000, 131072]))
    g = torch.Generator().manual_seed(seed)
    feats = []
    for (z, y, x) in dims:
        f = (torch.randn(1, C, z, y, x, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)
        feats.append(f.requires_grad_(True))
    b = torch.tensor(bound)
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.2 + b[:, 0] - 0.1 * (b[:, 1] - b[:, 0])
    k = int(rs.choice([0, n // 4, n // 2]))
    if k:
        x[:k] = x[:k] * 0.02 + b.mean(dim=1)                     # a tight cluster: very uneven tiles
    x[-1] = float("nan")
    ignore = [bool(rs.rand() < 0.2) and L > 1 for _ in range(L)]
    meta = ops.GridMeta.from_bound(bound, ignore_level=ignore)
    go = torch.randn(n, C * L, generator=g).to(DEV)

    def run():
        xd = x.to(DEV).requires_grad_(True)
        out = ops.encode(xd, feats, meta)
        grads = torch.autograd.grad(out, feats + [xd], go, allow_unused=True)
        return out.detach(), grads

    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", 16384)
    out_b, gb = run()
    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", None)
    out_a, ga = run()
    valid = ~torch.isnan(out_a).any(dim=1)
    assert torch.equal(out_b[valid], out_a[valid])
    for l, (a, c) in enumerate(zip(gb, ga)):
        if a is None or c is None:
            assert a is None and c is None
            continue
        a, c = torch.nan_to_num(a), torch.nan_to_num(c)
        assert a.shape == c.shape
        assert relerr(a, c) < 5e-5, (seed, l, dims, C, n)


@pytest.mark.parametrize("seed", range(4))
def test_levels_with_large_bricks_next_to_pulled_ones(seed, monkeypatch):
    """Levels whose tile bricks hold more than 8 vertices on an axis (ScanNet's 200 x 100 x 200 over 16 tiles) are not
    owned by the pull: in a binned batch they are scattered with float atomics while the levels next to them are pulled
    (the split of miso_grad_pull_levels).  Same gradients as the unbinned path, first and second order, with a tight
    cluster and points outside the bound."""
    from miso_amd import ops
    rs = np.random.RandomState(7000 + seed)
    C = int(rs.choice([4, 8]))
    big = [tuple(int(v) for v in rs.choice([100, 130, 144, 160, 200, 255, 256], size=3))]
    if seed == 0:
        big = [(200, 100, 200)]
    dims = [tuple(int(v) for v in rs.choice([8, 20, 40, 64], size=3))] + big
    L = len(dims)
    bmin = rs.uniform(-3, 0, size=3)
    bmax = bmin + rs.uniform(0.5, 6, size=3)
    bound = [[float(bmin[a]), float(bmax[a])] for a in range(3)]
    n = int(rs.choice([16384, 50000, 131072]))
    g = torch.Generator().manual_seed(seed)
    feats = []
    for (z, y, x) in dims:
        f = (torch.randn(1, C, z, y, x, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)
        feats.append(f.requires_grad_(True))
    b = torch.tensor(bound)
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.1 + b[:, 0] - 0.05 * (b[:, 1] - b[:, 0])
    k = [0, n // 2, n // 8, 0][seed % 4]
    if k:
        x[:k] = x[:k] * 0.03 + b.mean(dim=1)                     # thousands of points on a handful of tiles
    x[-1] = float("nan")
    meta = ops.GridMeta.from_bound(bound)
    go = torch.randn(n, C * L, generator=g).to(DEV)
    want = [torch.empty_like(f) for f in feats]
    mask = int(ops._lib.load().miso_grad_pull_levels(ops.C.byref(ops._fill_grid(feats, meta, want, data=False)),
                                                     ops.SortedBatch.TILES))
    assert mask == 0b01, "the small level is pulled, the large one is not"

    def run(second):
        xd = x.to(DEV).requires_grad_(True)
        out = ops.encode(xd, feats, meta)
        if not second:
            return torch.autograd.grad(out, feats, go)
        (gx,) = torch.autograd.grad(out, xd, go, create_graph=True)
        return torch.autograd.grad((torch.nan_to_num(gx) ** 2).sum(), feats)

    for second in (False, True):
        monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", 16384)
        gb = run(second)
        monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", None)      # unbinned: everything with float atomics
        ga = run(second)
        for l, (a, c) in enumerate(zip(gb, ga)):
            a, c = torch.nan_to_num(a), torch.nan_to_num(c)
            assert relerr(a, c) < 5e-5, (seed, second, l, dims, C, n)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MISO_STRESS_SEEDS", "10"))))
def test_binned_fused_matches_plain(seed):
    """Fused encode+decoder: binned forward / backward (incl. the loss-fused forward and grad_x) against
    the plain kernels over the covered (C, L, H) table and irregular grid shapes."""
    from miso_amd import ops
    rs = np.random.RandomState(5000 + seed)
    C, L, H = [(4, 1, 32), (4, 1, 64), (4, 2, 32), (4, 2, 64), (4, 3, 64), (4, 4, 64), (8, 1, 64), (8, 2, 64),
               (8, 3, 64), (8, 4, 64), (8, 3, 32)][seed % 11]
    dims = [tuple(int(v) for v in rs.choice([8, 16, 24, 32, 40, 64, 100, 128], size=3)) for _ in range(L)]
    bmin = rs.uniform(-2, 0, size=3)
    bmax = bmin + rs.uniform(1.0, 5.0, size=3)
    bound = [[float(bmin[a]), float(bmax[a])] for a in range(3)]
    n = int(rs.choice([1, 63, 4097, 70000]))
    g = torch.Generator().manual_seed(seed)
    feats = [(torch.randn(1, C, z, y, x, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             for (z, y, x) in dims]
    lin = [torch.nn.Linear(C * L, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(DEV) for l in lin], [l.bias.data.to(DEV) for l in lin])
    meta = ops.GridMeta.from_bound(bound)
    assert ops.sdf_fused_supported(feats, meta, pack)
    b = torch.tensor(bound)
    x = (torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.1 + b[:, 0] - 0.05 * (b[:, 1] - b[:, 0])).to(DEV)
    gs = (torch.randn(n, 1, generator=g) / max(n, 1)).to(DEV)
    sdf_a, mask_a = ops.sdf_fwd_raw(x, feats, meta, pack, True)
    gx_a, gr_a = ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_a, True, [True] * L)
    sb = ops.SortedBatch(n, DEV).sort(x, meta)
    sdf_b, mask_b = ops.sdf_fwd_raw(x, feats, meta, pack, True, sorted_batch=sb)
    gx_b, gr_b = ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_b, True, [True] * L, sorted_batch=sb, overwrite=True)
    assert torch.equal(sdf_a, sdf_b)
    if n:
        assert relerr(gx_b, gx_a) < 1e-5
    for a, c in zip(gr_b, gr_a):
        assert relerr(a, c) < 5e-5, (seed, C, L, H, dims, n)
    # loss-fused forward: same d loss / d sdf (in binned order) and loss as the separate loss kernel
    if n:
        aux = torch.stack([torch.randn(n, generator=g) * 0.1, (torch.rand(n, generator=g) > 0.2).float(),
                           (torch.rand(n, generator=g) > 0.7).float(), torch.rand(n, generator=g) + 0.5], dim=1).to(DEV)
        gsort = torch.empty(n, device=DEV)
        slots = torch.empty(ops._lib.LOSS_SLOTS, 2, device=DEV)
        ops.sdf_fwd_loss_raw(feats, meta, pack, sb, aux.contiguous(), mask_b, gsort, slots, "L1", 1.0, 0.3, 0.2)
        terms, gref = ops.mapping_loss_raw(sdf_a, aux[:, 0:1].contiguous(), aux[:, 1:2].contiguous(),
                                           aux[:, 2:3].contiguous(), aux[:, 3:4].contiguous(), "L1", 1.0, 0.3, 0.2)
        assert torch.allclose(slots.sum(0), terms, rtol=1e-5, atol=1e-7)
        assert torch.equal(gsort, gref.reshape(-1)[sb.perm.long()])


@pytest.mark.parametrize("shape", ["coarse_only", "cfg2"])
def test_heavy_tiles_are_cut_and_summed_exactly(shape, monkeypatch):
    """A batch that crowds 60 % of its points into two spots.  The matrix-core pull (grad_pull_mc.hip) works such blocks
    off in epochs (table and pool overflow); the vector kernels (MISO_PULL_MC=0) cut the over-full tiles into slices
    that a second launch adds.  Same gradients from both, as with the cut disabled (one wavefront per tile) up to fp32
    summation order, same as the atomic scatter, and the slice queue is left rewound."""
    from miso_amd import ops
    g = torch.Generator().manual_seed(5)
    if shape == "coarse_only":      # ScanNet's coarse level: 2-3 vertices per tile and axis
        dims, C, bound = [(20, 40, 40)], 4, [[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]]
    else:
        dims, C, bound = [(32,) * 3, (64,) * 3, (128,) * 3], 8, [[-1.0, 1.0]] * 3
    feats = [(torch.randn(1, C, *d, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             .requires_grad_(True) for d in dims]
    b = torch.tensor(bound)
    n = 150000
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) + b[:, 0]
    ext = (b[:, 1] - b[:, 0])
    x[: n * 3 // 10] = b.mean(1) + (torch.rand(n * 3 // 10, 3, generator=g) - 0.5) * ext * 0.05
    x[n * 3 // 10: n * 6 // 10] = b[:, 0] + ext * 0.3 + torch.randn(n * 3 // 10, 3, generator=g) * ext * 0.01
    meta = ops.GridMeta.from_bound(bound)
    go = torch.randn(n, C * len(dims), generator=g).to(DEV)
    queues = []
    orig_init = ops.SortedBatch.__init__

    def spy_init(self, *a, **k):
        orig_init(self, *a, **k)
        queues.append(self.pull_queue)

    monkeypatch.setattr(ops.SortedBatch, "__init__", spy_init)

    def run():
        xd = x.to(DEV)
        out = ops.encode(xd, feats, meta)
        return torch.autograd.grad(out, feats, go)

    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", 16384)
    g_mc = run()
    assert all(int(q.abs().sum()) == 0 for q in queues)                  # the matrix-core pull queues nothing
    monkeypatch.setenv("MISO_PULL_MC", "0")
    g_split = run()
    assert queues and all(int(q[:4].abs().sum()) == 0 for q in queues)   # header rewound after use
    assert any(int(q[4:].abs().sum()) > 0 for q in queues)               # and slices were indeed queued
    monkeypatch.setenv("MISO_PULL_NO_SPLIT", "1")
    g_whole = run()
    monkeypatch.delenv("MISO_PULL_NO_SPLIT")
    monkeypatch.delenv("MISO_PULL_MC")
    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", None)
    g_atomic = run()
    for m, a, w, c in zip(g_mc, g_split, g_whole, g_atomic):
        assert relerr(m, c) < 2e-5
        assert relerr(a, w) < 2e-5
        assert relerr(a, c) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("seed,C", [(0, 4), (1, 4), (2, 4), (3, 8)])
def test_dense_wave_scatter_matches_pull_and_atomics(seed, C, monkeypatch):
    """the matrix-core push of coarse levels under a crowd (the knob MISO_DENSE_MIN is read once by the
    library, so the crowd is real): a batch that averages >= 100 points per tile on a two-level ScanNet-like grid
    pushes both levels; the gradients equal the unsorted atomic path's to fp32 summation-order tolerance."""
    from miso_amd import ops
    torch.manual_seed(seed)
    dev = "cuda:0"
    H = 64
    # (X, Y, Z): coarse bricks <= 3 per axis at 16^3 tiles (<= 2 with 8 channels: the tiny-brick kernel's register budget)
    sizes = [(8, 5, 8), (40, 25, 40)] if C == 4 else [(8, 5, 8), (32, 20, 27)]
    feats = [(torch.randn(1, C, z, y, x, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
             for (x, y, z) in sizes]
    meta = ops.GridMeta.from_bound([[-2.0, 2.0], [-1.0, 1.5], [-2.0, 2.0]])
    lin = [torch.nn.Linear(2 * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(dev) for l in lin], [l.bias.data.to(dev) for l in lin])
    n = 100 * 4096 + 777                                  # >= 100 per tile on average
    x = torch.rand(n, 3, device=dev) * torch.tensor([1.6, 0.9, 1.2], device=dev) + torch.tensor([-0.9, -0.2, 0.1], device=dev)
    x[:50] = torch.rand(50, 3, device=dev) * 8 - 4        # some far outside the bound
    gs = torch.randn(n, 1, device=dev) / n
    import ctypes
    from miso_amd import _lib
    grid = ops._fill_grid(feats, meta, grads=[torch.zeros_like(f) for f in feats])
    # both levels (regions of <= 5 vertices per axis and tile) go through the matrix-core push
    assert _lib.load().miso_sdf_bwd_push_levels(ctypes.byref(grid), 16, n) == 0b11
    assert _lib.load().miso_sdf_bwd_push_levels(ctypes.byref(grid), 16, 64 * 4096) == 0
    sdf, mask = ops.sdf_fwd_raw(x, feats, meta, pack, True)
    want = [torch.zeros_like(f) for f in feats]
    ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask, False, [True, True], want)
    sb = ops.SortedBatch(n, dev).sort(x, meta)
    _, mask_s = ops.sdf_fwd_raw(x, feats, meta, pack, True, sorted_batch=sb)
    got = [torch.full_like(f, 7.0) for f in feats]        # overwrite mode must not depend on what was there
    ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_s, False, [True, True], got, sorted_batch=sb, overwrite=True)
    for a, b in zip(got, want):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-4 * scale + 1e-12      # thousands of fp32 terms per coarse vertex, in two orders


@pytest.mark.gpu
def test_captured_step_replayed_back_to_back_between_eager_launches():
    """The captured mapping step replayed 60 times with no host sync and eager launches in between (set_batch's copies,
    the loss sum, Adam), the way GridTrainer drives it.  Regression: the zero-fill of the atomically scattered levels
    was a hipMemsetAsync node, which under exactly this pattern filled with garbage now and then on ROCm 7.2 -- the
    gradients gained non-zeros all over the level and Adam's active-chunk flags went to 100 %.  It is a kernel now."""
    from miso_amd import ops
    from miso_amd.step import MappingStep
    torch.manual_seed(0)
    dev, C, H, n = "cuda:0", 4, 64, 540000
    sizes = [(40, 20, 40), (200, 100, 200)]                                  # ScanNet (cfg-3): push + atomics
    meta = ops.GridMeta.from_bound([[-10., 10.], [-5., 5.], [-10., 10.]])
    lin = [torch.nn.Linear(2 * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(dev) for l in lin], [l.bias.data.to(dev) for l in lin])
    x = (torch.rand(n, 3, device=dev) * 2 - 1) * torch.tensor([6.0, 2.5, 6.0], device=dev)
    tgt = torch.rand(n, 1, device=dev) * 0.2 - 0.1
    one, zero = torch.ones(n, 1, device=dev), torch.zeros(n, 1, device=dev)
    feats = [(torch.randn(1, C, z, y, xx, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
             for (xx, y, z) in sizes]
    st = MappingStep(feats, meta, pack, n, "L1", 1.0, 0.1, 0.15, use_graph=True, sort=True, keep_sdf=False)
    st.set_batch(x, tgt, one, zero, one)
    for _ in range(3):
        st.run()
    torch.cuda.synchronize()
    support = [(g != 0) for g in st.grads]
    m, v = [torch.zeros_like(f) for f in feats], [torch.zeros_like(f) for f in feats]
    act = [ops.adam_active_flags(f) for f in feats]
    for it in range(60):
        xw = x * 1.0
        st.set_batch(xw, tgt, one, zero, one)
        del xw
        st.run()
        tot = st.loss.sum()
        for f, g, mm, vv, a in zip(feats, st.grads, m, v, act):
            ops.adam_active_(f, g, mm, vv, a, it + 1, 1e-3, 0.9, 0.999, 1e-8, zero_grad=False, guard=tot.reshape(1))
    torch.cuda.synchronize()
    for g, s, a in zip(st.grads, support, act):
        assert torch.isfinite(g).all()
        assert not (g != 0)[~s].any()                   # nothing outside the vertices the batch can touch
        assert float(a.float().mean()) < 0.5            # and Adam still skips the empty part of the bound


@pytest.mark.gpu
def test_crowded_hint_bins_small_batches_and_pushes_the_coarse_level():
    """MappingStep(crowded=True) (what Mapper asks for: ray samples pile up on a few tiles) bins a 54 000-sample batch
    and pushes the coarse ScanNet level through the matrix cores although the average density is 13 samples a tile;
    same loss and gradients as the default step, which takes the unbinned atomic path at this size."""
    from miso_amd import ops, _lib
    from miso_amd.step import MappingStep
    torch.manual_seed(0)
    dev, C, H, n = "cuda:0", 4, 64, 54000
    sizes = [(40, 20, 40), (200, 100, 200)]
    meta = ops.GridMeta.from_bound([[-10., 10.], [-5., 5.], [-10., 10.]])
    lin = [torch.nn.Linear(2 * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(dev) for l in lin], [l.bias.data.to(dev) for l in lin])
    x = (torch.rand(n, 3, device=dev) * 2 - 1) * torch.tensor([6.0, 2.5, 6.0], device=dev)
    x[: n // 2] = x[: n // 2] * 0.05 + torch.tensor([1.0, 0.5, -2.0], device=dev)        # half of it on a handful of tiles
    tgt = torch.rand(n, 1, device=dev) * 0.2 - 0.1
    one, zero = torch.ones(n, 1, device=dev), torch.zeros(n, 1, device=dev)
    feats = [(torch.randn(1, C, z, y, xx, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
             for (xx, y, z) in sizes]
    res = []
    for crowded in (False, True):
        st = MappingStep(feats, meta, pack, n, "L1", 1.0, 0.1, 0.15, use_graph=False, keep_sdf=False, crowded=crowded)
        assert (st.sorted is not None) == crowded
        st.set_batch(x, tgt, one, zero, one)
        st.run()
        torch.cuda.synchronize()
        res.append((st.loss.clone(), [g.clone() for g in st.grads]))
        if crowded:
            assert st.meta.flags & _lib.F_CROWDED
            g = ops._fill_grid(feats, st.meta, st.grads, data=False)
            assert int(_lib.load().miso_sdf_bwd_push_levels(ops.C.byref(g), ops.SortedBatch.TILES, n)) == 0b01
            g = ops._fill_grid(feats, meta, st.grads, data=False)
            assert int(_lib.load().miso_sdf_bwd_push_levels(ops.C.byref(g), ops.SortedBatch.TILES, n)) == 0
    (la, ga), (lb, gb) = res
    assert torch.allclose(la.sum(), lb.sum(), rtol=1e-5, atol=1e-7)
    for a, c in zip(ga, gb):
        assert relerr(c, a) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_matrix_core_push_random_shapes_and_clusters(seed):
    """The push on random level sizes up to 48 per axis (a tile owns <= 3 vertices: regions of <= 5), 4 or 8
    channels, batches that are part uniform, part tight clusters (thousands of samples in one tile: runs of 512 samples
    inside one tile, tiles split between runs), part outside the bound -- against the unsorted atomic scatter."""
    from miso_amd import ops, _lib
    import ctypes
    gen = torch.Generator().manual_seed(100 + seed)
    dev = "cuda:0"
    C = 4 if seed % 2 == 0 else 8
    H = 64
    sizes = [tuple(int(v) for v in torch.randint(8, 49, (3,), generator=gen)) for _ in range(2)]
    feats = [(torch.randn(1, C, z, y, x, generator=gen) * 1e-2).to(dev).contiguous(memory_format=torch.channels_last_3d)
             for (x, y, z) in sizes]
    lo = torch.tensor([-3.0, -1.0, -2.0]) * (1 + seed * 0.3)
    hi = torch.tensor([2.0, 1.5, 4.0]) * (1 + seed * 0.3)
    meta = ops.GridMeta.from_bound([[float(a), float(b)] for a, b in zip(lo, hi)])
    lin = [torch.nn.Linear(2 * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(dev) for l in lin], [l.bias.data.to(dev) for l in lin])
    n = 100 * 4096 + int(torch.randint(0, 5000, (1,), generator=gen))
    x = torch.rand(n, 3, generator=gen) * (hi - lo) + lo
    k = n // 4
    ctr = torch.rand(3, 3, generator=gen) * (hi - lo) + lo
    for i in range(3):                                                   # three tight clusters
        x[i * k // 3:(i + 1) * k // 3] = ctr[i] + torch.randn((i + 1) * k // 3 - i * k // 3, 3, generator=gen) * 0.01 * (hi - lo)
    x[k:k + 300] = (torch.rand(300, 3, generator=gen) - 0.5) * 4 * (hi - lo) + 0.5 * (hi + lo)     # many outside
    x = x[torch.randperm(n, generator=gen)].to(dev)
    gs = (torch.randn(n, 1, generator=gen) / n).to(dev)
    grid = ops._fill_grid(feats, meta, grads=[torch.zeros_like(f) for f in feats])
    assert _lib.load().miso_sdf_bwd_push_levels(ctypes.byref(grid), 16, n) == 0b11
    sdf, mask = ops.sdf_fwd_raw(x, feats, meta, pack, True)
    want = [torch.zeros_like(f) for f in feats]
    ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask, False, [True, True], want)
    sb = ops.SortedBatch(n, dev).sort(x, meta)
    _, mask_s = ops.sdf_fwd_raw(x, feats, meta, pack, True, sorted_batch=sb)
    got = [torch.full_like(f, -3.0) for f in feats]
    ops.sdf_bwd_raw(x, feats, meta, pack, gs, mask_s, False, [True, True], got, sorted_batch=sb, overwrite=True)
    for a, b in zip(got, want):
        scale = b.abs().max().item()
        assert scale > 0 and (a - b).abs().max().item() <= 2e-4 * scale + 1e-12


@pytest.mark.parametrize("shape", ["scannet", "uneven"])
def test_per_axis_binning_lets_the_pull_own_fine_levels(shape, monkeypatch):
    """ScanNet's submap (configs/rgbd/scannet.yaml:23-24: 40 x 20 x 40 and 200 x 100 x 200 vertices) under the 16-tile
    binning puts 12.5 vertices of the fine level on a tile and axis -- beyond what the pull owns, so that level was
    scattered with float atomics.  With MISO_STEP_TILES=auto MappingStep bins such grids per axis (ops.choose_tiles: (25, 16, 25)), every level
    goes through the matrix-core pull, and the gradients equal those of the 16-tile step (atomics for the fine level,
    push for the coarse one) to fp32 summation order, crowded batch and points outside the bound included."""
    from miso_amd import ops
    from miso_amd.step import MappingStep
    g = torch.Generator().manual_seed(17)
    if shape == "scannet":
        C, dims, bound, want = 4, [(40, 20, 40), (200, 100, 200)], [[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]], (25, 16, 25)
    else:     # sizes no tile count divides, 8 channels, three levels
        C, dims, bound, want = 8, [(23, 31, 17), (91, 130, 70), (182, 250, 141)], [[-1.0, 2.0], [0.0, 4.0], [-3.0, 0.5]], \
            (23, 32, 18)
    feats = [(torch.randn(1, C, z, y, x, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             for (x, y, z) in dims]
    F = C * len(dims)
    lin = [torch.nn.Linear(F, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 1)]
    pack = ops.DecoderPack([m.weight.detach().to(DEV) for m in lin], [m.bias.detach().to(DEV) for m in lin])
    meta = ops.GridMeta.from_bound(bound)
    assert ops.choose_tiles(feats) == ops.pack_tiles(want)
    n = 140000
    b = torch.tensor(bound)
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.06 + b[:, 0] - 0.03 * (b[:, 1] - b[:, 0])
    x[: n // 2] = b.mean(1) + (torch.rand(n // 2, 3, generator=g) - 0.5) * (b[:, 1] - b[:, 0]) * 0.08       # a crowd
    aux = torch.stack((torch.randn(n, generator=g) * 0.1, (torch.rand(n, generator=g) > 0.1).float(),
                       (torch.rand(n, generator=g) > 0.6).float(), torch.rand(n, generator=g) + 0.5), dim=1).to(DEV)
    out = {}
    for tiles in ("auto", "16"):
        monkeypatch.setenv("MISO_STEP_TILES", tiles)
        st = MappingStep(feats, meta, pack, n, "L1", 1.0, 0.1, 0.15, keep_sdf=False, sort=True)
        grads = [torch.full_like(f, 3.0) for f in feats]
        scat = ops.sdf_train_scattered_levels(feats, meta, grads, tiles=st.tiles)
        assert (scat == 0) == (tiles == "auto"), (tiles, scat)         # per-axis: every level pulled
        st.set_batch(x.to(DEV), aux[:, 0:1], aux[:, 1:2], aux[:, 2:3], aux[:, 3:4])
        st.run(); st.run()
        torch.cuda.synchronize()
        out[tiles] = (st.loss.clone(), [g_.clone() for g_ in st.grads])
    assert (out["auto"][0] - out["16"][0]).abs().max().item() <= 2e-6 * out["16"][0].abs().max().item()
    for a, c in zip(out["auto"][1], out["16"][1]):
        assert relerr(a, c) < 2e-5


@pytest.mark.parametrize("seed", range(6))
def test_matrix_core_pull_epochs_and_halved_ranges(seed, monkeypatch):
    """The matrix-core pull's rarely taken paths on ordinary batches: with MISO_MC_SMALL its survivor table holds 192
    entries and its pair pool 640, so EVERY block is worked off in several epochs (first stores, later ones
    read-add-store) and ranges whose pairs would overflow the pool are halved.  Same gradients as the full-size
    kernel and as the atomic scatter; random level shapes (sizes no tile count divides among them), C = 4 / 8, 1-4 levels,
    an ignored level, accumulate mode (grad += through autograd's second call), clusters and points outside the bound."""
    from miso_amd import ops
    rs = np.random.RandomState(4200 + seed)
    L = int(rs.randint(1, 5))
    C = int(rs.choice([4, 8]))
    dims = [tuple(int(v) for v in rs.choice([16, 20, 24, 33, 48, 64, 100, 128], size=3)) for _ in range(L)]
    bound = [[-1.0, 1.5], [0.0, 2.0], [-2.0, 0.0]]
    n = int(rs.choice([20011, 70000]))
    g = torch.Generator().manual_seed(seed)
    feats = [(torch.randn(1, C, z, y, x, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             .requires_grad_(True) for (z, y, x) in dims]
    b = torch.tensor(bound)
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.1 + b[:, 0] - 0.05 * (b[:, 1] - b[:, 0])
    if seed % 2:
        x[: n // 3] = x[: n // 3] * 0.03 + b.mean(dim=1)
    ignore = [L > 1 and l == 1 and seed % 3 == 0 for l in range(L)]
    meta = ops.GridMeta.from_bound(bound, ignore_level=ignore)
    go = torch.randn(n, C * L, generator=g).to(DEV)

    def run():
        out = ops.encode(x.to(DEV), feats, meta)
        return torch.autograd.grad(out, feats, go, allow_unused=True)

    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", 16384)
    g_full = run()
    monkeypatch.setenv("MISO_MC_SMALL", "1")
    g_small = run()
    monkeypatch.delenv("MISO_MC_SMALL")
    monkeypatch.setattr(ops, "ENCODE_PULL_MIN_POINTS", None)
    g_atomic = run()
    for l, (a, s_, c) in enumerate(zip(g_full, g_small, g_atomic)):
        if c is None:
            assert a is None and s_ is None
            continue
        assert relerr(s_, c) < 5e-5 and relerr(a, c) < 5e-5, (seed, l, dims, C)
