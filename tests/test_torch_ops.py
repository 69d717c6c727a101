"""miso_amd.torch_ops: the C-ABI entry points as ``torch.library`` operators.  Without a GPU: the eight
operators exist with the documented schemas, their fake kernels give the shapes, and there is no CPU kernel behind them.
On the GPU: each equals the ``miso_amd.ops`` call it wraps (same C entry, same arguments: bit for bit), autograd reaches
second order through encode_fwd -> encode_bwd -> encode_bwd2, and ``torch.library.opcheck`` passes."""
import pytest
import torch

from miso_amd import torch_ops as T

BOUND = [-1.0, -0.5, 0.0, 1.0, 1.5, 2.0]


def test_the_eight_operators_are_registered_with_tensor_only_schemas():
    for name in T.OPS:
        schema = str(getattr(torch.ops.miso, name).default._schema)
        assert schema.startswith(f"miso::{name}("), schema
    s = str(torch.ops.miso.adam_dense.default._schema)
    assert "Tensor(a0!) param" in s and "Tensor(a3!) exp_avg_sq" in s          # declared in-place
    assert "float[] bound" in str(torch.ops.miso.encode_fwd.default._schema)


def test_fake_kernels_give_the_shapes_without_a_gpu():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(100, 3, device="cuda")
        fs = [torch.empty(1, 4, 8, 8, 8, device="cuda"), torch.empty(1, 4, 16, 16, 16, device="cuda")]
        out = torch.ops.miso.encode_fwd(x, fs, BOUND, 0, 0)
        assert out.shape == (100, 8)
        gx, g0, g1 = torch.ops.miso.encode_bwd(out, x, fs, BOUND, 0, 0, True, [True, False])
        assert gx.shape == (100, 3) and g0.shape == fs[0].shape and g1.numel() == 0
        gg = torch.ops.miso.encode_bwd2(out, x, fs, BOUND, 0, 0, gx, [g0, g1], True, [True, True])
        assert gg[0].shape == (100, 8) and gg[1].shape == (100, 3) and gg[2].shape == fs[0].shape
        ws = [torch.empty(64, 8, device="cuda"), torch.empty(64, 64, device="cuda"), torch.empty(1, 64, device="cuda")]
        bs = [torch.empty(64, device="cuda"), torch.empty(64, device="cuda"), torch.empty(1, device="cuda")]
        sdf, mask = torch.ops.miso.encode_decode_fwd(x, fs, ws, bs, BOUND, 0, 0, True)
        assert sdf.shape == (100, 1) and mask.dtype == torch.int32 and mask.numel() == 128 * 4
        H, g, s = torch.ops.miso.lm_normal_eq(x, torch.empty(3, 3, device="cuda"), x, sdf, sdf, 2, 0.1)
        assert H.shape == (6, 6) and g.shape == (6, 1) and s.shape == ()


def test_there_is_no_cpu_kernel():
    x = torch.zeros(4, 3)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.miso.encode_fwd(x, [torch.zeros(1, 4, 4, 4, 4)], BOUND, 0, 0)


def _grids(C, sizes, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(1, C, s, s, s, generator=g) * 0.1).cuda().contiguous(memory_format=torch.channels_last_3d)
            for s in sizes]


def _points(n, seed):
    g = torch.Generator().manual_seed(seed)
    b = torch.tensor(BOUND).view(2, 3)
    return (torch.rand(n, 3, generator=g) * (b[1] - b[0]) * 1.1 + b[0] - 0.05 * (b[1] - b[0])).cuda()


@pytest.mark.gpu
def test_encode_operator_equals_the_autograd_function_to_second_order():
    from miso_amd import ops
    meta = ops.GridMeta(tuple(BOUND[:3]), tuple(BOUND[3:]), 0, 0)
    fs_a = [f.requires_grad_() for f in _grids(4, (8, 20), 1)]
    fs_b = [f.detach().clone().requires_grad_() for f in fs_a]
    xa = _points(3000, 2).requires_grad_()
    xb = xa.detach().clone().requires_grad_()
    out_a = torch.ops.miso.encode_fwd(xa, fs_a, *T.meta_args(meta))
    out_b = ops.encode(xb, fs_b, meta)
    assert torch.equal(out_a, out_b)
    w = torch.randn_like(out_a)
    ga = torch.autograd.grad((out_a * w).sum(), [xa] + fs_a, create_graph=True)
    gb = torch.autograd.grad((out_b * w).sum(), [xb] + fs_b, create_graph=True)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=0, atol=2e-6 * float(b.detach().abs().max()))      # atomics: summation order
    # an eikonal-like term on the coordinate gradient, differentiated again
    ea = ((ga[0].norm(dim=1) - 1.0) ** 2).mean()
    eb = ((gb[0].norm(dim=1) - 1.0) ** 2).mean()
    ha = torch.autograd.grad(ea, [xa] + fs_a)
    hb = torch.autograd.grad(eb, [xb] + fs_b)
    for a, b in zip(ha, hb):
        assert torch.allclose(a, b, rtol=0, atol=2e-5 * float(b.detach().abs().max()))


@pytest.mark.gpu
def test_encode_decode_operator_equals_sdf_fused():
    from miso_amd import ops
    meta = ops.GridMeta(tuple(BOUND[:3]), tuple(BOUND[3:]), 0, 0)
    fs_a = [f.requires_grad_() for f in _grids(4, (8, 20), 3)]
    fs_b = [f.detach().clone().requires_grad_() for f in fs_a]
    torch.manual_seed(0)
    lin = [torch.nn.Linear(8, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 1)]
    ws = [m.weight.detach().cuda() for m in lin]
    bs = [m.bias.detach().cuda() for m in lin]
    x = _points(5000, 4)
    sdf_a, mask = torch.ops.miso.encode_decode_fwd(x, fs_a, ws, bs, *T.meta_args(meta), True)
    sdf_b = ops.sdf_fused(x, fs_b, meta, ops.DecoderPack(ws, bs))
    assert torch.equal(sdf_a, sdf_b) and mask.numel() == ((5000 + 63) // 64) * 64 * 4
    w = torch.randn_like(sdf_a)
    ga = torch.autograd.grad((sdf_a * w).sum(), fs_a)
    gb = torch.autograd.grad((sdf_b * w).sum(), fs_b)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=0, atol=2e-6 * float(b.abs().max()))


@pytest.mark.gpu
def test_alignment_tracking_and_adam_operators_equal_their_ops_calls():
    from miso_amd import ops
    meta = ops.GridMeta(tuple(BOUND[:3]), tuple(BOUND[3:]), 0, 0)
    fs = _grids(4, (8, 20), 5)
    x = _points(4000, 6)
    src = torch.randn(4000, 8, generator=torch.Generator().manual_seed(7)).cuda()
    R = torch.eye(3).cuda()
    t = torch.tensor([[0.05], [0.02], [-0.03]]).cuda()
    pose = torch.cat((R.reshape(9), t.reshape(3), R.reshape(9), torch.zeros(3).cuda()))
    out = torch.ops.miso.pair_latent_fwd_bwd(pose, x, src, fs, *T.meta_args(meta), 2)
    loss = ops.pair_latent(R, t, R, torch.zeros(3, 1).cuda(), x, src, fs, meta, "L2")
    assert out.dtype == torch.float64 and out.shape == (24,)
    assert torch.allclose((out[0] / (out[1].clamp(min=1.0) * 8)).float(), loss, rtol=1e-6, atol=0)
    gw = torch.nn.functional.normalize(torch.randn(4000, 3, generator=torch.Generator().manual_seed(8)), dim=1).cuda()
    sp, sg = torch.randn(4000, 1).cuda() * 0.1, torch.randn(4000, 1).cuda() * 0.1
    H1, g1, s1 = torch.ops.miso.lm_normal_eq(x, R, gw, sp, sg, 2, 0.1)
    H2, g2, s2 = ops.lm_normal_eq(x, R, gw, sp, sg, "L2", 0.1)
    assert torch.allclose(H1, H2, rtol=1e-5) and torch.allclose(g1, g2, rtol=1e-5, atol=1e-6) and torch.allclose(s1, s2, rtol=1e-5)
    p1 = torch.randn(1000, device="cuda")
    p2, g = p1.clone(), torch.randn(1000, device="cuda")
    m1, v1, m2, v2 = (torch.zeros(1000, device="cuda") for _ in range(4))
    torch.ops.miso.adam_dense(p1, g.clone(), m1, v1, 1, 1e-2, 0.9, 0.999, 1e-8, False)
    ops.adam_dense_(p2, g.clone(), m2, v2, 1, 1e-2, 0.9, 0.999, 1e-8, False)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)


@pytest.mark.gpu
def test_opcheck_encode_and_adam():
    fs = [f.requires_grad_() for f in _grids(4, (8,), 9)]
    x = _points(257, 10).requires_grad_()
    checks = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.miso.encode_fwd.default, (x, fs, BOUND, 0, 0), test_utils=checks)
    p, g, m, v = (torch.randn(100, device="cuda") for _ in range(4))
    torch.library.opcheck(torch.ops.miso.adam_dense.default, (p, g, m, v.abs(), 1, 1e-2, 0.9, 0.999, 1e-8, True),
                          test_utils=("test_schema", "test_faketensor"))


def test_decoder_pack_cache_keys_on_layout_and_stays_small():
    """the operator layer keeps one DecoderPack per decoder.  Two views of ONE buffer (same base pointer,
    another shape) are different decoders and must not share a pack; the cache is a small LRU (the oldest entry goes,
    nothing is cleared wholesale in the middle of a forward / backward pair).  Host logic: no kernel runs here."""
    buf = torch.zeros(64 * 64)
    w_a = [buf[: 64 * 8].view(64, 8), buf.view(64, 64), buf[:64].view(1, 64)]
    w_b = [buf[: 32 * 16].view(32, 16), buf[: 32 * 32].view(32, 32), buf[:32].view(1, 32)]       # the same storage, reshaped
    none = [torch.empty(0)] * 3
    T._packs.clear()
    p_a, p_b = T._pack(w_a, none), T._pack(w_b, none)
    assert p_a is not p_b and p_a is T._pack(w_a, none) and len(T._packs) == 2
    for i in range(3 * T._PACKS_MAX):
        T._pack([torch.zeros(8, 8), torch.zeros(1, 8)], [torch.empty(0)] * 2)
        T._pack(w_a, none)                                             # in use: never the oldest
    assert len(T._packs) <= T._PACKS_MAX and T._pack(w_a, none) is p_a
