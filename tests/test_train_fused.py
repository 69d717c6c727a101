"""sdf_train_kernel (miso_sdf_train_sorted: forward + mapping loss + decoder backward in ONE launch, then the pull) against
the two-launch form it replaces (miso_sdf_fwd_sorted_loss + miso_sdf_bwd_sorted): same SDF and loss bit for bit (the
forward is the same instruction sequence), the same d-feat rows -- hence gradients equal up to the pull's summation
order -- over the fused shape table, both losses, free-space term, invalid / NaN-labelled rows, padded batches (n_live),
a crowded batch whose coarse level goes through the matrix-core push, and an ignored level.  Plus: a level the pull
cannot own is scattered from the train kernel itself, and unbinned batches (miso_sdf_train) against their two launches.
The two-launch form itself is pinned to the reference goldens and the CPU oracle elsewhere (test_hip_parity.py,
test_config_shapes.py); test_train_kernel_vs_cpu_oracle below compares the one-launch kernel with the oracle DIRECTLY
(stock ATen grid_sample per level, cat, nn.Linear chain, the reference's two loss terms, autograd), every shape of the
table, the scattering variant and the unbinned form included."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R          # checker only (CPU restatement of the reference, pinned by tests/golden)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # (C, level sizes (cubic), hidden)
    (8, (32, 64, 128), 64),      # cfg-2
    (4, (16, 80), 64),           # ScanNet-like ratio 5, both levels pullable
    (4, (48,), 32),
    (8, (16, 32, 64, 128), 64),
    (8, (32, 48, 64), 32),
    (4, (16, 32, 48, 64), 64),
]


def _setup(C, sizes, H, n, seed, crowded=False, ignore=None, bound=None, want_lin=False):
    from miso_amd import ops
    g = torch.Generator().manual_seed(seed)
    feats = [(torch.randn(1, C, s, s, s, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last_3d)
             for s in sizes]
    F = C * len(sizes)
    lin = [torch.nn.Linear(F, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    torch.manual_seed(seed)
    for m in lin:
        torch.nn.init.normal_(m.weight, std=0.3)
    pack = ops.DecoderPack([m.weight.detach().to(DEV) for m in lin], [m.bias.detach().to(DEV) for m in lin])
    bound = bound or [[-1.0, 1.0], [-0.5, 1.5], [0.0, 2.0]]
    meta = ops.GridMeta.from_bound(bound, ignore_level=ignore)
    if crowded:
        import dataclasses
        meta = dataclasses.replace(meta, flags=meta.flags | ops._lib.F_CROWDED)
    b = torch.tensor(bound)
    x = torch.rand(n, 3, generator=g) * (b[:, 1] - b[:, 0]) * 1.1 + b[:, 0] - 0.05 * (b[:, 1] - b[:, 0])
    if crowded:
        x[: n // 2] = x[: n // 2] * 0.05 + b.mean(dim=1)
    aux = torch.stack((torch.randn(n, generator=g) * 0.1, (torch.rand(n, generator=g) > 0.15).float(),
                       (torch.rand(n, generator=g) > 0.6).float(), torch.rand(n, generator=g) + 0.5), dim=1)
    if n > 7:
        aux[7, 0] = float("nan")
    out = (feats, meta, pack, x.to(DEV).contiguous(), aux.to(DEV).contiguous())
    return out + (lin,) if want_lin else out


def _both(feats, meta, pack, x, aux, lt, ws, wf, td, n_live=None, need=None):
    from miso_amd import ops
    n, L = x.shape[0], len(feats)
    need = need or [True] * L
    sb = ops.SortedBatch(n, DEV).sort(x, meta)
    mask = torch.empty(((n + 63) // 64) * 64 * ops.sdf_mask_words(pack), device=DEV, dtype=torch.int32)
    gpred = torch.empty(n, 1, device=DEV)
    s2, sdf2 = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=DEV), torch.empty(n, 1, device=DEV)
    g2 = [torch.full_like(f, 7.0) if nd else None for f, nd in zip(feats, need)]
    ops.sdf_fwd_loss_raw(feats, meta, pack, sb, aux, mask, gpred, s2, lt, ws, wf, td, sdf_out=sdf2, n_live=n_live)
    ops.sdf_bwd_raw(x, feats, meta, pack, gpred, mask, False, need, g2, sorted_batch=sb, overwrite=True, gsdf_sorted=True)
    s1, sdf1 = torch.zeros_like(s2), torch.empty_like(sdf2)
    g1 = [torch.full_like(f, -3.0) if nd else None for f, nd in zip(feats, need)]
    assert ops.sdf_train_supported(feats, meta, g1)
    ops.sdf_train_raw(feats, meta, pack, sb, aux, s1, g1, lt, ws, wf, td, sdf_out=sdf1, n_live=n_live)
    torch.cuda.synchronize()
    return (s1, sdf1, g1), (s2, sdf2, g2)


def _check(a, b):
    (s1, sdf1, g1), (s2, sdf2, g2) = a, b
    # the per-workgroup loss slots: the two forms group the wavefronts' sums differently (eight per workgroup in the
    # one-launch kernel, four in the forward kernel) -- the totals agree to the rounding of that regrouping
    t1, t2 = s1.double().sum(0), s2.double().sum(0)
    assert torch.equal(torch.isnan(t1), torch.isnan(t2))
    ok = ~torch.isnan(t2)
    assert torch.allclose(t1[ok], t2[ok], rtol=2e-6, atol=1e-12), (t1, t2)
    assert torch.equal(sdf1, sdf2)
    for x1, x2 in zip(g1, g2):
        if x1 is None:
            assert x2 is None
            continue
        fin = torch.isfinite(x2)
        assert torch.equal(fin, torch.isfinite(x1))
        scale = x2[fin].abs().max().item()
        assert (x1[fin] - x2[fin]).abs().max().item() <= 2e-6 * scale      # the pull's summation order only


@pytest.mark.parametrize("shape", range(len(SHAPES)))
@pytest.mark.parametrize("lt", ["L1", "L2"])
def test_train_kernel_equals_forward_plus_backward(shape, lt):
    C, sizes, H = SHAPES[shape]
    feats, meta, pack, x, aux = _setup(C, sizes, H, 70001, seed=shape)
    aux[7, 0] = 0.0                                   # (NaN labels: below)
    a, b = _both(feats, meta, pack, x, aux, lt, 1.0, 0.1, 0.15)
    _check(a, b)
    assert all(float(g.abs().max()) > 0 for g in a[2])


def test_train_kernel_nan_label_padded_batch_and_subset_of_levels():
    from miso_amd import ops
    feats, meta, pack, x, aux = _setup(8, (32, 64, 128), 64, 66000, seed=11)
    a, b = _both(feats, meta, pack, x, aux, "L1", 1.0, 0.0, 0.0)
    assert torch.isnan(a[0].sum()) and torch.isnan(b[0].sum())        # the NaN label poisons the loss of both alike
    aux[7, 0] = 0.5
    aux[60000:] = 0.0                                                  # neutral padding rows
    live = torch.tensor([60000], device=DEV, dtype=torch.int32)
    _check(*_both(feats, meta, pack, x, aux, "L2", 0.7, 0.2, 0.1, n_live=live))
    _check(*_both(feats, meta, pack, x, aux, "L1", 1.0, 0.1, 0.1, need=[False, True, True]))
    # an ignored level: its features do not enter, its gradient is written as zeros by both forms
    feats, meta, pack, x, aux = _setup(8, (32, 64, 128), 64, 66000, seed=12, ignore=[False, True, False])
    aux[7, 0] = 0.0
    a, b = _both(feats, meta, pack, x, aux, "L1", 1.0, 0.1, 0.1)
    _check(a, b)
    assert float(a[2][1].abs().max()) == 0.0


def test_train_kernel_with_the_matrix_core_push_on_a_crowded_batch():
    from miso_amd import ops
    feats, meta, pack, x, aux = _setup(4, (40, 80), 64, 120000, seed=21, crowded=True,
                                       bound=[[-10.0, 10.0], [-10.0, 10.0], [-10.0, 10.0]])
    aux[7, 0] = 0.0
    grads = [torch.empty_like(f) for f in feats]
    assert ops.sdf_bwd_scattered_levels(feats, meta, grads, x.shape[0]) & 1        # level 0 is pushed
    _check(*_both(feats, meta, pack, x, aux, "L1", 1.0, 0.1, 0.15))


def test_levels_the_pull_cannot_own_are_scattered_from_the_train_kernel():
    """A level with bricks beyond the pull's reach (200 vertices over 16 tiles: 12.5 per tile and axis; cfg-3's fine
    level) no longer forces the two launches: sdf_train_kernel<.., SCAT> scatters it with float atomics itself, the
    other level still goes through the d-feat rows.  Against the two-launch form: SDF bit for bit, the loss to the regrouping of the slots, the pulled
    level to the pull's summation order, the scattered one to the order of the atomics."""
    from miso_amd import ops
    from miso_amd.step import MappingStep
    feats, meta, pack, x, aux = _setup(4, (40, 200), 64, 70000, seed=31)
    aux[7, 0] = 0.0
    grads = [torch.empty_like(f) for f in feats]
    assert ops.sdf_train_supported(feats, meta, grads) and ops.sdf_train_scattered_levels(feats, meta, grads) == 2
    (s1, sdf1, g1), (s2, sdf2, g2) = _both(feats, meta, pack, x, aux, "L1", 1.0, 0.1, 0.15)
    # (the loss slots: eight wavefronts per workgroup in the one-launch kernel where its records fit, four in the forward
    # kernel -- the totals agree to the rounding of that regrouping, as in _check)
    assert torch.allclose(s1.double().sum(0), s2.double().sum(0), rtol=2e-6, atol=1e-12) and torch.equal(sdf1, sdf2)
    assert (g1[0] - g2[0]).abs().max().item() <= 2e-6 * g2[0].abs().max().item()
    assert (g1[1] - g2[1]).abs().max().item() <= 2e-5 * g2[1].abs().max().item() and float(g2[1].abs().max()) > 0
    # only the scattered level wanted: nothing goes through the workspace at all
    a, b = _both(feats, meta, pack, x, aux, "L2", 1.0, 0.0, 0.0, need=[False, True])
    assert a[2][0] is None and (a[2][1] - b[2][1]).abs().max().item() <= 2e-5 * b[2][1].abs().max().item()
    step = MappingStep(feats, meta, pack, x.shape[0], "L1", 1.0, 0.1, 0.15, keep_sdf=False)
    assert step.sorted is not None and step._fused_train()
    step.set_batch(x, aux[:, 0:1], aux[:, 1:2], aux[:, 2:3], aux[:, 3:4])
    step.run()
    torch.cuda.synchronize()
    assert torch.isfinite(step.loss).all()
    for a_, b_ in zip(step.grads, g2):
        assert (a_ - b_).abs().max().item() <= 2e-5 * b_.abs().max().item()


@pytest.mark.parametrize("n", [1, 65, 3000, 40000])
def test_rotated_scattering_kernel_on_small_and_ragged_binned_batches(n):
    """sdf_train_kernel<.., SCAT> with narrow feature rows runs eight wavefronts per workgroup with its loop rotated (the next
    chunk's gathers in front of this chunk's atomics, two blocks of cell records): batches with fewer chunks than
    wavefronts (a wavefront without a chunk), one ragged chunk, and several chunks per wavefront -- against the two-launch
    form, the scattered level to the order of the atomics."""
    feats, meta, pack, x, aux = _setup(4, (40, 200), 64, n, seed=50 + n)
    if n > 7:
        aux[7, 0] = 0.0
    (s1, sdf1, g1), (s2, sdf2, g2) = _both(feats, meta, pack, x, aux, "L1", 1.0, 0.1, 0.15)
    assert torch.allclose(s1.double().sum(0), s2.double().sum(0), rtol=2e-6, atol=1e-12) and torch.equal(sdf1, sdf2)
    for a, b in zip(g1, g2):
        scale = max(b.abs().max().item(), 1e-30)
        assert (a - b).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("shape", [(4, (16, 80), 64), (8, (32, 64, 128), 64), (4, (48,), 32), (4, (16, 32, 48, 64), 64)])
@pytest.mark.parametrize("n", [1, 65, 3000])
def test_unbinned_train_kernel_equals_forward_plus_backward(shape, n):
    """miso_sdf_train (an unbinned batch: forward + loss + decoder backward + the atomic scatter of every level in one
    launch) against miso_sdf_fwd_loss + miso_sdf_bwd: SDF bit for bit, loss to the order of the per-workgroup sums,
    gradients (added to what the buffers held) to the order of the float atomics; touched flags alike."""
    from miso_amd import ops
    C, sizes, H = shape
    feats, meta, pack, x, aux = _setup(C, sizes, H, n, seed=n + len(sizes))
    if n > 7:
        aux[7, 0] = 0.0
    else:
        aux[:, 0] = 0.1
    L = len(feats)
    mask = torch.empty(((n + 63) // 64) * 64 * ops.sdf_mask_words(pack), device=DEV, dtype=torch.int32)
    gpred = torch.empty(n, 1, device=DEV)
    s2, sdf2 = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=DEV), torch.empty(n, 1, device=DEV)
    g2 = [torch.full_like(f, 0.25) for f in feats]
    t2 = [ops.adam_active_flags(f) for f in feats]
    ops.sdf_fwd_loss_unsorted_raw(x, feats, meta, pack, aux, mask, gpred, s2, "L1", 1.0, 0.1, 0.15, sdf_out=sdf2)
    ops.sdf_bwd_raw(x, feats, meta, pack, gpred, mask, False, [True] * L, g2, touched=t2)
    s1, sdf1 = torch.ones_like(s2), torch.empty_like(sdf2)
    g1 = [torch.full_like(f, 0.25) for f in feats]
    t1 = [ops.adam_active_flags(f) for f in feats]
    ops.sdf_train_unsorted_raw(x, feats, meta, pack, aux, s1, g1, "L1", 1.0, 0.1, 0.15, sdf_out=sdf1, touched=t1)
    torch.cuda.synchronize()
    assert torch.equal(sdf1, sdf2)
    assert (s1.sum(0) - s2.sum(0)).abs().max().item() <= 2e-6 * max(s2.sum(0).abs().max().item(), 1e-12)
    for a, b, ta, tb in zip(g1, g2, t1, t2):
        scale = (b - 0.25).abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-7         # (0.25 + tiny: one ulp of 0.25 is 3e-8)
        assert torch.equal(ta, tb)


def test_stream_launches_and_graph_replays_give_the_same_step():
    """use_graph=None picks by batch size (MappingStep.STREAM_MIN_POINTS); both launch modes run the same kernels."""
    from miso_amd.step import MappingStep
    n = MappingStep.STREAM_MIN_POINTS
    feats, meta, pack, x, aux = _setup(8, (32, 64, 128), 64, n, seed=41)
    aux[7, 0] = 0.0
    out = {}
    for mode in (None, True, False):
        st = MappingStep(feats, meta, pack, n, "L1", 1.0, 0.1, 0.15, keep_sdf=False, use_graph=mode)
        assert st._use_graph == (mode is True)                       # None at this size: stream launches
        st.set_batch(x, aux[:, 0:1], aux[:, 1:2], aux[:, 2:3], aux[:, 3:4])
        for _ in range(3):
            st.run()
        torch.cuda.synchronize()
        out[mode] = (st.loss.clone(), [g.clone() for g in st.grads])
    small = MappingStep(feats, meta, pack, n // 4, "L1", 1.0, 0.1, 0.15, keep_sdf=False)
    assert small._use_graph
    for mode in (None, False):
        # (not bit for bit: the order of the points inside a tile run depends on the sort's LDS atomics, and with it the
        # fp32 order of the loss and gradient sums)
        assert (out[mode][0] - out[True][0]).abs().max().item() <= 2e-6 * out[True][0].abs().max().item()
        for a, b in zip(out[mode][1], out[True][1]):
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()


def _oracle(feats, meta_bound, lin, x, aux, lt, ws, wf, td, n_live=None):
    """Loss terms and grid gradients of the reference on the host: GridNet.forward (grid_net.py:306-325) ->
    miso_loss_regression + miso_loss_free_space (loss.py:594-635, :668-700), weights as MisoLossMapping applies them."""
    fc = [f.detach().cpu().contiguous().clone().requires_grad_(True) for f in feats]
    w = [m.weight.detach().clone() for m in lin]
    b = [m.bias.detach().clone() for m in lin]
    xc, ac = x.cpu(), aux.cpu()
    if n_live is not None:                    # the reference sees the exact-size batch
        xc, ac = xc[:n_live], ac[:n_live]
    pred = R.sdf_stock(fc, torch.tensor(meta_bound), xc, w, b)
    t_sdf = ws * R.miso_loss_regression(pred, ac[:, 0:1], ac[:, 1:2], ac[:, 3:4], lt)
    t_fs = wf * R.miso_loss_free_space(pred, ac[:, 0:1], ac[:, 2:3], td)
    grads = torch.autograd.grad(t_sdf + t_fs, fc)
    return pred.detach(), torch.stack((t_sdf.detach(), t_fs.detach())), grads


def _check_vs_oracle(sdf, loss, grads, ref, n_live=None):
    pred, terms, gref = ref
    if sdf is not None:
        d = (sdf.cpu()[: pred.shape[0]] - pred).abs()
        assert d.max().item() <= 1e-5, d.max().item()
    assert (loss.cpu() - terms).abs().max().item() <= 1e-6 * max(1.0, terms.abs().max().item())
    for a, r in zip(grads, gref):
        a = a.cpu()
        scale = r.abs().max().item()
        assert scale > 0
        # (a sample whose ReLU pre-activation lies within fp32 rounding of zero may be gated differently by two
        # implementations -- ~1 in 6 000 samples, tests/test_config_shapes.py counts them at full size -- and moves the eight
        # vertices around it by one sample's share: the Euclidean norm and the fraction of entries bound the rest)
        assert ((a - r).double().norm() / r.double().norm()).item() < 5e-4
        # ... at most a handful of tie samples' 8 corners x C channels beyond 1e-4 of the largest entry
        assert int(((a - r).abs() > 1e-4 * scale).sum()) <= max(2e-4 * a.numel(), 8 * a.shape[1] * 6)


_BOUND = [[-1.0, 1.0], [-0.5, 1.5], [0.0, 2.0]]


def _setup_lin(C, sizes, H, n, seed, **kw):
    """_setup plus the nn.Linear modules its DecoderPack was made from (the oracle needs the weights)."""
    return _setup(C, sizes, H, n, seed, want_lin=True, **kw)


@pytest.mark.parametrize("lt", ["L1", "L2"])
@pytest.mark.parametrize("shape", list(range(len(SHAPES))) + ["scat"])
def test_train_kernel_vs_cpu_oracle(shape, lt):
    """miso_sdf_train_sorted (sdf_train_kernel: forward + mapping loss + decoder backward in one launch, then the pull /
    push; `scat`: a 200-vertex level the pull cannot own is scattered from the kernel, sdf_train_kernel<..,SCAT>) against
    the oracle: SDF to 1e-5, both loss terms to 1e-6, every level's gradient to 1e-4 of its largest entry (ReLU ties
    apart), with invalid rows, free-space rows, per-sample weights and points outside the bound."""
    from miso_amd import ops
    C, sizes, H = (4, (40, 200), 64) if shape == "scat" else SHAPES[shape]
    n = 70001
    feats, meta, pack, x, aux, lin = _setup_lin(C, sizes, H, n, seed=100 + (7 if shape == "scat" else shape))
    aux[7, 0] = 0.05
    ws, wf, td = 1.0, 0.2, 0.15
    sb = ops.SortedBatch(n, DEV).sort(x, meta)
    slots, sdf = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=DEV), torch.empty(n, 1, device=DEV)
    grads = [torch.full_like(f, -3.0) for f in feats]
    if shape == "scat":
        assert ops.sdf_train_scattered_levels(feats, meta, grads) == 2
        grads[1].zero_()                      # a scattered level is added to: the caller clears it (MappingStep: Adam does)
    ops.sdf_train_raw(feats, meta, pack, sb, aux, slots, grads, lt, ws, wf, td, sdf_out=sdf)
    torch.cuda.synchronize()
    _check_vs_oracle(sdf, slots.sum(0), grads, _oracle(feats, _BOUND, lin, x, aux, lt, ws, wf, td))


def test_train_kernel_padded_batch_vs_cpu_oracle():
    """A fixed-capacity batch whose rows past n_live are neutral padding (miso_sample_rays leaves them so): the means divide
    by the live count, as the reference's do over its exact-size batch."""
    from miso_amd import ops
    C, sizes, H = SHAPES[0]
    n, live = 66000, 60000
    feats, meta, pack, x, aux, lin = _setup_lin(C, sizes, H, n, seed=131)
    aux[7, 0] = 0.05
    aux[live:] = 0.0
    x[live:] = x[:n - live]                   # padding rows are parked on live samples
    sb = ops.SortedBatch(n, DEV).sort(x, meta)
    slots = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=DEV)
    grads = [torch.empty_like(f) for f in feats]
    nl = torch.tensor([live], device=DEV, dtype=torch.int32)
    ops.sdf_train_raw(feats, meta, pack, sb, aux, slots, grads, "L1", 0.7, 0.2, 0.1, n_live=nl)
    torch.cuda.synchronize()
    _check_vs_oracle(None, slots.sum(0), grads, _oracle(feats, _BOUND, lin, x, aux, "L1", 0.7, 0.2, 0.1, n_live=live))


@pytest.mark.parametrize("n", [65, 3000])
def test_unbinned_train_kernel_vs_cpu_oracle(n):
    """miso_sdf_train (unbinned: every level scattered with float atomics from the one launch) against the oracle."""
    from miso_amd import ops
    C, sizes, H = SHAPES[1]
    feats, meta, pack, x, aux, lin = _setup_lin(C, sizes, H, n, seed=150 + n)
    aux[7, 0] = 0.05
    slots, sdf = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=DEV), torch.empty(n, 1, device=DEV)
    grads = [torch.zeros_like(f) for f in feats]
    ops.sdf_train_unsorted_raw(x, feats, meta, pack, aux, slots, grads, "L2", 1.0, 0.3, 0.15, sdf_out=sdf)
    torch.cuda.synchronize()
    pred, terms, gref = _oracle(feats, _BOUND, lin, x, aux, "L2", 1.0, 0.3, 0.15)
    assert (sdf.cpu() - pred).abs().max().item() <= 1e-5
    assert (slots.sum(0).cpu() - terms).abs().max().item() <= 1e-6 * max(1.0, terms.abs().max().item())
    for a, r in zip(grads, gref):
        assert (a.cpu() - r).abs().max().item() <= 1e-4 * r.abs().max().item()


def test_train_kernel_takes_the_index_from_xn_when_the_batch_has_no_perm_array():
    """miso_sort_points leaves the original index in xn_sorted[p].w; miso_sdf_train_sorted accepts a batch without perm[]
    (MappingStep's fused step: one scattered store per point less in the sort).  Same results bit for bit as with
    perm[]; every other entry point refuses such a batch."""
    from miso_amd import ops
    feats, meta, pack, x, aux = _setup(8, (32, 64, 128), 64, 70001, seed=21)
    aux[7, 0] = 0.0
    n = x.shape[0]
    sb_p = ops.SortedBatch(n, DEV).sort(x, meta)
    sb_n = ops.SortedBatch(n, DEV, need_perm=False).sort(x, meta)
    assert sb_n.perm is None
    assert torch.equal(sb_p.xn_sorted[:, 3].view(torch.int32), sb_p.perm)          # the index rides in .w either way
    assert torch.equal(sb_p.xn_sorted[:, :3][torch.argsort(sb_p.perm)], sb_n.xn_sorted[:, :3][torch.argsort(
        sb_n.xn_sorted[:, 3].view(torch.int32))])                                   # (orders inside a tile may differ)
    outs = []
    for sb in (sb_p, sb_n):
        slots, sdf = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=DEV), torch.full((n, 1), 9.0, device=DEV)
        grads = [torch.full_like(f, -3.0) for f in feats]
        ops.sdf_train_raw(feats, meta, pack, sb, aux, slots, grads, "L1", 1.0, 0.1, 0.15, sdf_out=sdf)
        outs.append((slots, sdf, grads))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][1], outs[1][1])                                      # sdf in the caller's order
    assert torch.allclose(outs[0][0].double().sum(0), outs[1][0].double().sum(0), rtol=2e-6, atol=0)
    for a, b in zip(outs[0][2], outs[1][2]):
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()           # the pull's summation order only
    mask = torch.empty(((n + 63) // 64) * 64 * ops.sdf_mask_words(pack), device=DEV, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        ops.sdf_fwd_raw(x, feats, meta, pack, True, mask=mask, sorted_batch=sb_n)
    with pytest.raises(RuntimeError):
        ops.encode_fwd_raw(x, feats, meta, sorted_batch=sb_n)


@pytest.mark.parametrize("shape", [(4, (16, 80), 64), (8, (32, 64, 128), 64)])
def test_small_batch_32_point_trips_equal_64_point_trips(shape):
    """an unbinned batch of at most 65 536 samples runs sdf_train_kernel in 32-point trips (HALF: lanes 32-63
    mirror 0-31, one point tile per wavefront).  Against the 64-point form of the same kernel on the same batch
    (MISO_F_FULL_TRIPS): SDF bit for bit, loss slots to the order of the per-workgroup sums, gradients to the order of the
    float atomics, touched flags alike."""
    from miso_amd import ops
    import dataclasses
    C, sizes, H = shape
    n = 6144
    feats, meta, pack, x, aux = _setup(C, sizes, H, n, seed=3)
    aux[7, 0] = 0.0                                          # (_setup plants a NaN label for the guard tests)
    full = dataclasses.replace(meta, flags=meta.flags | ops._lib.F_FULL_TRIPS)
    outs = []
    for m in (meta, full):
        s_, sdf = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=DEV), torch.empty(n, 1, device=DEV)
        g = [torch.zeros_like(f) for f in feats]
        t = [ops.adam_active_flags(f) for f in feats]
        ops.sdf_train_unsorted_raw(x, feats, m, pack, aux, s_, g, "L1", 1.0, 0.1, 0.15, sdf_out=sdf, touched=t)
        torch.cuda.synchronize()
        outs.append((s_, sdf, g, t))
    (s1, sdf1, g1, t1), (s2, sdf2, g2, t2) = outs
    assert torch.equal(sdf1, sdf2)
    assert (s1.sum(0) - s2.sum(0)).abs().max().item() <= 2e-6 * max(s2.sum(0).abs().max().item(), 1e-12)
    for a, b, ta, tb in zip(g1, g2, t1, t2):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-9
        assert torch.equal(ta, tb)
