"""The fused atlas query (round 6: miso_atlas_sdf_fwd, csrc/atlas.hip) -- GridAtlas.query_feature / forward and the
lattice evaluation behind save_mesh as ONE launch -- against
  * the reference's own outputs (tests/golden/atlas.npz, formats.npz: written by the reference, tools/make_goldens.py),
  * today's op-by-op loop (the reference's per-submap structure, grid_opt/models/grid_atlas.py:374-399) on the
    ScanNet-shaped 8-submap atlas of bench.py, points inside several / one / no submap,
  * itself: the lattice form (points generated in the kernel) equals the point-list form bit for bit, the exact-fp32
    decoder form stays within 1e-5 of the split one.
fp32 tolerances as the reference tests of the mirror: features 2e-6, SDF 1e-5."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from test_grid_opt_mirror import G, T, close, make_atlas

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _loop(atlas, x):
    """the per-submap loop (what runs when autograd is on): evaluated with autograd on, detached"""
    with torch.enable_grad():
        return atlas(x).detach(), atlas.query_feature(x).detach()


def test_fused_atlas_query_vs_reference_goldens():
    g = G("atlas")
    atlas = make_atlas(DEV)
    xw = T(gc.atlas_world_points()).to(DEV)
    with torch.no_grad():
        got = atlas._fused_query(xw, want_sdf=True, want_feats=True)
        assert got is not None, "the fused query was not taken"
        close(got[1], T(g["query_feature"]), 0, 2e-6)
        close(got[0], T(g["forward"]), 0, 1e-5)
        close(atlas.query_feature(xw), T(g["query_feature"]), 0, 2e-6)
        close(atlas(xw), T(g["forward"]), 0, 1e-5)
    # the whole-module pickle written by the reference (demo/build_submaps.py:141)
    import miso_amd.compat  # noqa: F401
    f = G("formats")
    ref_atlas = torch.load(os.path.join(gc.GOLDEN_DIR, "ref_atlas.pth"), weights_only=False, map_location="cpu").to(DEV)
    xr = T(f["atlas_x"]).to(DEV)
    with torch.no_grad():
        assert ref_atlas._fused_query(xr) is not None
        close(ref_atlas(xr), T(f["atlas_forward"]), 0, 1e-5)
        close(ref_atlas.query_feature(xr), T(f["atlas_query_feature"]), 0, 2e-6)


@pytest.fixture(scope="module")
def scannet8():
    import bench
    return bench.scannet_atlas(DEV, 8)


def test_fused_atlas_query_vs_the_loop_on_eight_scannet_submaps(scannet8):
    atlas = scannet8
    gb = atlas.global_bound(device="cpu")
    gen = torch.Generator().manual_seed(11)
    n = 200000
    lo, hi = gb[:, 0] - 1.0, gb[:, 1] + 1.0                  # a margin: points inside no submap
    x = (lo + (hi - lo) * torch.rand(n, 3, generator=gen)).to(DEV)
    sdf_l, feat_l = _loop(atlas, x)
    with torch.no_grad():
        sdf_f, feat_f = atlas(x), atlas.query_feature(x)
        assert atlas._fused_query(x) is not None
    # the three populations are all present
    cnt = torch.zeros(n, device=DEV)
    for s in atlas.active_submaps:
        import miso_amd.grid_opt.utils.utils_geometry as ug
        R, t = atlas.updated_submap_pose(s)
        cnt += ug.coords_in_bound(ug.transfrom_points_from(x, R, t), atlas.get_submap(s).bound.to(DEV)).float().view(-1)
    assert int((cnt == 0).sum()) > 1000 and int((cnt == 1).sum()) > 1000 and int((cnt >= 2).sum()) > 1000
    close(feat_f, feat_l, 0, 2e-6)
    close(sdf_f, sdf_l, 0, 1e-5)
    # a point inside no submap decodes the zero row
    zero = atlas.submaps[0].decoder(torch.zeros(1, feat_l.shape[1], device=DEV)).detach()
    assert (sdf_f[cnt == 0] - zero).abs().max().item() <= 1e-6
    # exact fp32 decoder chains behind the switch
    from miso_amd import ops
    with torch.no_grad(), ops.exact_fp32():
        sdf_e = atlas(x)
    if not os.environ.get("MISO_EXACT_F32"):       # (MISO_EXACT_F32=1 runs the whole suite on the exact chains: both are then the same launch)
        assert not torch.equal(sdf_e, sdf_f)
    close(sdf_e, sdf_l, 0, 1e-5)


def test_lattice_form_equals_the_point_list_form_and_feeds_save_mesh(scannet8, tmp_path):
    import miso_amd.grid_opt.utils.utils_sdf as US
    atlas = scannet8
    gb = atlas.global_bound(device="cpu")
    res = (37, 29, 41)                                       # ragged: the last chunk is partial
    axes = [torch.linspace(float(gb[a, 0]), float(gb[a, 1]), res[a]) for a in range(3)]
    vol = atlas.sdf_on_lattice(*[a.to(DEV) for a in axes])
    assert vol is not None and tuple(vol.shape) == res
    xx, yy, zz = torch.meshgrid(*[a.to(DEV) for a in axes], indexing="ij")
    pts = torch.stack((xx, yy, zz), dim=-1).reshape(-1, 3)
    with torch.no_grad():
        assert torch.equal(vol.reshape(-1, 1), atlas(pts))
    close(vol.reshape(-1, 1), _loop(atlas, pts)[0], 0, 1e-5)
    # extract_fields: the lattice path and the slab-by-slab query_func path give the same volume
    r = 48
    q = lambda p: atlas(p)                                   # noqa: E731
    with torch.no_grad():
        a = US.extract_fields_device(gb[:, 0], gb[:, 1], r, q, DEV, lattice_func=atlas.sdf_on_lattice)
        b = US.extract_fields_device(gb[:, 0], gb[:, 1], r, q, DEV)
    assert torch.equal(a, b)
    # (random features: shift the decoder's output bias so that the field has a zero level set to mesh)
    last = atlas.submaps[0].decoder.linears()[-1]
    with torch.no_grad():
        shift = a.mean().reshape(1)
        last.bias -= shift
    try:
        mesh = US.save_mesh(atlas, gb, save_path=str(tmp_path / "m" / "atlas.ply"), resolution=r, device=DEV)
    finally:
        with torch.no_grad():
            last.bias += shift
    assert mesh.vertices.shape[0] > 100 and mesh.triangles.shape[0] > 100
    assert np.isfinite(mesh.vertices).all()


def test_fused_query_is_not_taken_where_autograd_is_needed(scannet8, tmp_path):
    atlas = scannet8
    x = torch.zeros(16, 3, device=DEV)
    with torch.no_grad():
        before = atlas(x)
    torch.save(atlas, tmp_path / "atlas.pth")               # the query's device plan stays out of the pickle
    again = torch.load(tmp_path / "atlas.pth", weights_only=False)
    with torch.no_grad():
        assert torch.equal(again(x), before)
    # a pose correction that moves invalidates the cached pose table
    with torch.no_grad():
        atlas.translation_corrections[0] += 0.25
        moved = atlas(x)
        atlas.translation_corrections[0] -= 0.25
        assert not torch.equal(moved, before) and torch.equal(atlas(x), before)
    assert atlas._fused_query(x) is None                     # autograd on: the differentiable loop serves the call
    with torch.no_grad():
        assert atlas._fused_query(x) is not None
        assert atlas._fused_query(x.cpu()) is None           # host tensors: no device path


@pytest.mark.parametrize("C,L,H", [(8, 3, 64), (4, 1, 32), (8, 4, 64)])
def test_fused_atlas_query_other_shapes_vs_the_loop(C, L, H):
    """Three overlapping submaps of the cfg-2 / cfg-1 / cfg-5 channel and level counts (different poses, one of them far
    rotated), random world points incl. a margin outside every bound: features 2e-6, SDF 1e-5 against the per-submap loop;
    a ragged point count (the last 64-point chunk is partial)."""
    from miso_amd.grid_opt.models.grid_atlas import GridAtlas
    import math
    cfg = {"name": "grid_net", "spatial_dim": 3,
           "decoder": {"type": "mlp", "hidden_dim": H, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                       "fix": True, "pretrained_model": None},
           "grid": {"type": "regular", "feature_dim": C, "init_stddev": 3e-2, "bound": [[-1.0, 1.0], [-0.5, 0.75], [-1.0, 1.0]],
                    "base_cell_size": 0.25, "per_level_scale": 2, "n_levels": L},
           "pose": {"optimize": False, "num_poses": 1}}
    torch.manual_seed(3)
    atlas = GridAtlas(cfg, device=DEV)
    lb = torch.tensor(cfg["grid"]["bound"])
    for s, (tx, ang) in enumerate(((0.0, 0.0), (1.2, 0.3), (-0.8, -1.1))):
        Rz = torch.tensor([[math.cos(ang), -math.sin(ang), 0.0], [math.sin(ang), math.cos(ang), 0.0], [0.0, 0.0, 1.0]])
        atlas.add_submap(lb, Rz, torch.tensor([[tx], [0.1 * s], [-0.2 * s]]), num_poses=1)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
    atlas.to(DEV)
    gb = atlas.global_bound(device="cpu").detach()
    gen = torch.Generator().manual_seed(L)
    n = 50001
    x = ((gb[:, 0] - 0.3) + (gb[:, 1] - gb[:, 0] + 0.6) * torch.rand(n, 3, generator=gen)).to(DEV)
    sdf_l, feat_l = _loop(atlas, x)
    with torch.no_grad():
        got = atlas._fused_query(x, want_sdf=True, want_feats=True)
    assert got is not None
    close(got[1], feat_l, 0, 2e-6)
    close(got[0], sdf_l, 0, 1e-5)


def test_one_submap_without_the_bound_test_is_gridnet_forward():
    """MISO_F_ATLAS_NO_BOUND: the atlas kernel with ONE submap, an identity pose row and no bound test gives GridNet.forward on
    a lattice (a margin outside the bound included: zeros padding decides) bit for bit, an ignored level included.
    (Measured as the field of save_mesh(submap, ...) at 256^3: 1.85 ms against 1.75 ms for slabs of 4 M meshgrid points
    through sdf_fwd_kernel -- a single submap's slabs were never the bottleneck, so GridNet keeps the slab path.)"""
    from miso_amd import ops
    from test_grid_opt_mirror import make_gridnet
    case = gc.CASES["cfg2"]
    net = make_gridnet(case, DEV)
    b = net.bound.detach().cpu()
    res = (33, 20, 47)
    axes = [torch.linspace(float(b[a, 0]) - 0.1, float(b[a, 1]) + 0.1, res[a]).to(DEV) for a in range(3)]
    xx, yy, zz = torch.meshgrid(*axes, indexing="ij")
    pts = torch.stack((xx, yy, zz), dim=-1).reshape(-1, 3)
    ident = torch.tensor([[1., 0., 0., 0., 1., 0., 0., 0., 1., 0., 0., 0.]], device=DEV)
    q = ops.AtlasQuery()
    feats = [g.feature.detach() for g in net.features]
    for ignore in (None, 1):
        if ignore is not None:
            net.ignore_level(ignore)
        meta = net.features[0].grid_meta(net.ignore_level_)
        with torch.no_grad():
            vol, _ = q([feats], [meta], ident, net._fused_decoder(), axes=tuple(axes), no_bound=True)
            assert torch.equal(vol, net(pts))
            bounded, _ = q([feats], [meta], ident, net._fused_decoder(), axes=tuple(axes))
            assert not torch.equal(bounded, vol)             # with the test on, the margin decodes the zero row
