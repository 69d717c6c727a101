"""Marching cubes (the step after the path): the oracle against the third-party goldens
(scikit-image, tests/golden/mcubes.npz), the compiled case table against the oracle's derivation, and the HIP
sweeps against the oracle -- vertices and triangle indices bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import mcubes_ref as M

NAMES = ("sphere", "torus", "gyroid", "noise")
DEV = "cuda:0"


def G():
    return np.load(gc.golden_path("mcubes"))


def lexsorted(v):
    return v[np.lexsort((v[:, 2], v[:, 1], v[:, 0]))]


def test_case_table_properties():
    tab, cnt = M.case_table()
    assert tab.shape == (256, 15) and cnt.max() == 5 and cnt[0] == 0 and cnt[255] == 0
    for c in range(256):
        e = tab[c][tab[c] >= 0]
        assert len(e) == 3 * cnt[c]
        # every crossing edge of the case is used, and only crossing edges
        crossing = {k for k in range(12) if ((c >> M.EDGE_ENDS[k, 0]) & 1) != ((c >> M.EDGE_ENDS[k, 1]) & 1)}
        assert set(e.tolist()) == crossing
        # the complementary case covers the same edges
        assert set(tab[255 - c][tab[255 - c] >= 0].tolist()) == crossing


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_third_party_golden(name):
    g = G()
    u, iso = g[f"{name}_u"], float(g[f"{name}_iso"])
    v, f = M.marching_cubes(u, iso)
    assert v.dtype == np.float32 and f.dtype == np.int64 and f.max() == len(v) - 1
    gv = g[f"{name}_verts"]
    assert v.shape == gv.shape
    assert np.abs(lexsorted(v) - gv).max() <= 4e-6        # same edges crossed, same linear interpolation
    cu, cd = M.edge_manifold_counts(f)
    assert cu.max() == 2 and cd.max() == 1                # manifold, consistently oriented (open only at the volume border)
    area, vol = M.mesh_area_volume(v, f)
    if name != "noise":                                   # ambiguous cells: the two tables join them differently
        assert len(f) == int(g[f"{name}_faces"])
        assert abs(area - float(g[f"{name}_area"])) <= 1e-3 * float(g[f"{name}_area"])
    if name in ("sphere", "torus"):                       # closed surfaces
        assert cu.min() == 2 and len(v) - len(cu) + len(f) == (2 if name == "sphere" else 0)
        assert vol < 0                                    # raw normals point to the u < iso side
        assert abs(-vol - float(g[f"{name}_volume"])) <= 2e-3 * float(g[f"{name}_volume"])


def test_oracle_analytic_sphere_and_empty():
    n = 40
    grid = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).astype(np.float32)
    u = np.linalg.norm(grid - 19.3, axis=-1) - 12.7
    v, f = M.marching_cubes(u, 0.0)
    assert np.abs(np.linalg.norm(v - 19.3, axis=1) - 12.7).max() < 0.03          # chord error of linear interpolation
    area, vol = M.mesh_area_volume(v, f)
    assert abs(area / (4 * np.pi * 12.7 ** 2) - 1) < 5e-3 and abs(-vol / (4 / 3 * np.pi * 12.7 ** 3) - 1) < 5e-3
    v, f = M.marching_cubes(u, 1e3)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    v, f = M.marching_cubes(u[:1], 0.0)                                          # no cells along x
    assert v.shape == (0, 3) and f.shape == (0, 3)


def test_compiled_case_table_equals_derivation():
    from miso_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_int8 * 4096)()
    assert lib.miso_mc_case_table(buf) == 0
    got = np.frombuffer(buf, dtype=np.int8).reshape(256, 16)
    tab, cnt = M.case_table()
    assert np.array_equal(got[:, :15], tab) and np.array_equal(got[:, 15], cnt.astype(np.int8))
    assert lib.miso_mc_case_table(None) == _lib.E_BADARG
    assert lib.miso_mc_words(0, 4, 4) == -1 and lib.miso_mc_words(2048, 2048, 2048) == -1
    assert lib.miso_mc_words(1, 4, 4) == 4 and lib.miso_mc_words(3, 3, 65) == 18 and lib.miso_mc_words(256, 256, 256) == 1 << 18
    # per row and chunk of 64 samples: one sign word, three vertex words, two work-list slots
    assert lib.miso_mc_workspace_bytes(3, 3, 65) == 18 * 40
    assert lib.miso_mc_classify(None, 4, 4, 4, 0.0, None, None, None) == _lib.E_BADARG
    assert lib.miso_mc_emit(4, 4, 4, None, None, 0, 0, None, None) == _lib.E_BADARG


def test_ply_roundtrip_and_normals(tmp_path):
    from miso_amd.grid_opt.utils import utils_sdf as US
    g = G()
    v, f = M.marching_cubes(g["sphere_u"], 0.0)
    mesh = US.TriangleMesh(v, f[:, [2, 1, 0]])
    mesh.apply_transform(np.array([[0, -1, 0, 1.0], [1, 0, 0, 2.0], [0, 0, 1, 3.0], [0, 0, 0, 1]]))
    mesh.compute_vertex_normals()
    c = mesh.vertices.mean(0)
    out = ((mesh.vertices - c) * mesh.vertex_normals).sum(1)
    assert (out > 0).all()                                 # flipped faces: normals point out of the sphere
    p = tmp_path / "m.ply"
    mesh.export_ply(str(p))
    back = US.read_ply(str(p))
    assert np.array_equal(back.triangles, mesh.triangles) and np.allclose(back.vertices, mesh.vertices, atol=1e-5)
    assert open(p, "rb").read(3) == b"ply"


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_marching_cubes_equals_oracle(name):
    from miso_amd import ops
    g = G()
    u, iso = g[f"{name}_u"], float(g[f"{name}_iso"])
    v, f = ops.marching_cubes(torch.from_numpy(u).to(DEV), iso)
    rv, rf = M.marching_cubes(u, iso)
    assert np.array_equal(f.cpu().numpy(), rf)
    assert np.array_equal(v.cpu().numpy(), rv)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 2, 2), (3, 2, 5), (17, 16, 18), (33, 9, 70), (1, 8, 8), (5, 1, 7)])
def test_hip_marching_cubes_ragged_shapes(shape):
    from miso_amd import ops
    rs = np.random.RandomState(sum(shape))
    u = rs.standard_normal(shape).astype(np.float32)
    v, f = ops.marching_cubes(torch.from_numpy(u).to(DEV), 0.05)
    rv, rf = M.marching_cubes(u, 0.05)
    assert np.array_equal(f.cpu().numpy(), rf) and np.array_equal(v.cpu().numpy(), rv)
    v, f = ops.marching_cubes(torch.from_numpy(u).to(DEV), 100.0)               # nothing crosses
    assert v.shape == (0, 3) and f.shape == (0, 3)


@pytest.mark.gpu
def test_hip_marching_cubes_full_size_properties():
    """256^3 sphere + ripple: closed, consistently oriented, Euler characteristic 2, volume of the sphere."""
    from miso_amd import ops
    n = 256
    ax = torch.arange(n, device=DEV, dtype=torch.float32)
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    r = torch.sqrt((x - 127.3) ** 2 + (y - 126.1) ** 2 + (z - 128.9) ** 2)
    u = r - 90.0 + 0.8 * torch.sin(0.21 * x) * torch.cos(0.17 * y)
    v, f = ops.marching_cubes(u, 0.0)
    vn, fn = v.cpu().numpy(), f.cpu().numpy()
    cu, cd = M.edge_manifold_counts(fn)
    assert cu.min() == 2 and cu.max() == 2 and cd.max() == 1
    assert len(vn) - len(cu) + len(fn) == 2
    _, vol = M.mesh_area_volume(vn, fn)
    assert abs(-vol / (4 / 3 * np.pi * 90.0 ** 3) - 1) < 2e-3
    # every vertex lies on a lattice edge, at the zero of the linear interpolant
    frac = vn - np.floor(vn)
    assert ((frac > 0).sum(1) <= 1).all()


@pytest.mark.gpu
def test_save_mesh_of_a_gridnet(tmp_path):
    """save_mesh end to end: fused forward over the lattice -> HIP marching cubes -> PLY; equals the oracle run on
    the same volume, mapped to metres like the reference (utils_sdf.py:96-100)."""
    import test_grid_opt_mirror as tm
    from miso_amd.grid_opt.utils import utils_sdf as US
    case = gc.CASES["small"]
    net = tm.make_gridnet(case, DEV)
    bound = torch.tensor(case["bound"], dtype=torch.float32)
    res = 48
    with torch.no_grad():
        u = US.extract_fields(bound[:, 0], bound[:, 1], res, lambda p: net(p.to(DEV)), device=DEV)
    iso = float(np.median(u))                              # a random decoder's field need not cross zero
    verts, tris = US.extract_geometry(bound[:, 0], bound[:, 1], res, iso, lambda p: net(p.to(DEV)), device=DEV)
    rv, rf = M.marching_cubes(u, iso)
    lo, hi = bound[:, 0].numpy(), bound[:, 1].numpy()
    assert np.array_equal(tris, rf)
    assert np.allclose(verts, rv.astype(np.float64) / (res - 1.0) * (hi - lo) + lo, atol=1e-12)

    class Shifted(torch.nn.Module):
        def forward(self, p):
            return net(p) - iso

    T = torch.eye(4)
    T[:3, 3] = torch.tensor([1.0, -2.0, 0.5])
    mesh = US.save_mesh(Shifted(), bound, save_path=str(tmp_path / "sub" / "m.ply"), resolution=res, device=DEV,
                        transform=T)
    back = US.read_ply(str(tmp_path / "sub" / "m.ply"))
    assert len(back.triangles) == len(rf) and np.array_equal(back.triangles, mesh.triangles)
    assert mesh.vertex_normals.shape == mesh.vertices.shape
