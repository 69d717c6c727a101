"""GPU parity of the sample-generation row: miso_sample_rays through the C ABI against the CPU
oracle (oracle.rgbd_sdf_samples, itself pinned against the reference's PosedSdfRgbd.getitem_sdf in
tests/test_oracle_golden.py) and against the committed golden rows.

Tolerances: depths, SDF and world points agree to a few fp32 ulp (the kernel follows the reference op by op with
FMA contraction off; only sums of three products may associate differently).  The world -> keyframe change
subtracts two terms of the size of the pose translation, so coords_frame is compared at 4 ulp of that size.
The truncation labels are discrete: rows whose |sdf| sits within 1e-6 of the band edge may fall on either side.
"""
import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def random_frames(seed, B, H, W, hole=0.2, nan_normals=0.1, nan_depth=3):
    rs = np.random.RandomState(seed)
    depth = rs.uniform(0.4, 6.0, (B, H, W)).astype(np.float32)
    depth[rs.uniform(0, 1, (B, H, W)) < hole] = 0.0
    for _ in range(nan_depth):
        depth[rs.randint(B), rs.randint(H), rs.randint(W)] = np.nan
    normals = rs.standard_normal((B, H, W, 3)).astype(np.float32)
    normals[rs.uniform(0, 1, (B, H, W)) < nan_normals] = np.nan
    Rm = np.stack([gc.rodrigues(rs.uniform(-1.5, 1.5, 3)) for _ in range(B)]).astype(np.float32)
    t = rs.uniform(-5.0, 5.0, (B, 3, 1)).astype(np.float32)
    Tm = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    Tm[:, :3, :3] = Rm
    Tm[:, :3, 3:] = t
    return T(depth), T(normals), T(Rm), T(t), T(Tm)


def run_both(depth, normals, Rm, t, Tm, intr, pb, ph, pw, u, g, knobs, frame_ids=None, rays_per_frame=0,
             explicit_b=True):
    from miso_amd import ops
    inputs, gt, extra = R.rgbd_sdf_samples(depth, Tm, Rm, t, intr, pb, ph, pw, u, g, normals=normals,
                                           frame_ids=frame_ids, **knobs)
    d = lambda a: None if a is None else a.to(DEV)
    out = ops.sample_rays(d(depth), d(Tm), d(Rm), d(t), intr, d(ph), d(pw), d(u), d(g), normals=d(normals),
                          frame_ids=frame_ids, pix_b=d(pb) if explicit_b else None, rays_per_frame=rays_per_frame,
                          keep_world=True, **knobs)
    return inputs, gt, extra, out


def compare(inputs, gt, extra, out, knobs, t_scale):
    counts = out.counts.cpu().tolist()
    S = knobs["n_strat"] + knobs["n_surf"]
    assert counts == [extra["n_first"], extra["n_kept"], extra["n_kept"] * S, 0]
    n = extra["n_kept"] * S
    assert out.rows() == n
    aux = out.aux.cpu()
    torch.testing.assert_close(out.z_vals.cpu()[:n], extra["z_vals"].reshape(-1), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out.pc_world.cpu()[:n], extra["pc_world"], rtol=1e-6, atol=2e-6)
    torch.testing.assert_close(aux[:n, 0:1], gt["sdf"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out.coords_frame.cpu()[:n], inputs["coords_frame"], rtol=0,
                               atol=4 * 1.2e-7 * max(t_scale, 8.0))
    assert torch.equal(out.sample_frame_ids.cpu()[:n, None], inputs["sample_frame_ids"])
    edge = ((gt["sdf"].abs() - knobs["trunc_dist"]).abs() < 1e-6)[:, 0]
    assert torch.equal(aux[:n, 1][~edge] > 0, gt["sdf_valid"][:, 0][~edge])
    assert torch.equal(aux[:n, 2][~edge], gt["sdf_signs"][:, 0][~edge])
    assert torch.equal(aux[:n, 3], torch.ones(n))
    # neutral padding behind the live rows: zero labels, parked on live positions (or the origin when nothing lives)
    assert aux[n:].abs().sum().item() == 0.0
    pad = out.coords_frame.cpu()[n:]
    assert not torch.isnan(pad).any()
    if n and pad.shape[0]:
        live = out.coords_frame.cpu()[:n]
        probe = pad[:: max(1, pad.shape[0] // 64)]
        assert bool((probe[:, None, :] == live[None, :, :]).all(-1).any(1).all())


@pytest.mark.parametrize("tag", ["all", "sel"])
def test_sample_rays_vs_reference_golden(tag):
    """The committed rows of the reference's own getitem_sdf (tests/golden/samples.npz)."""
    from miso_amd import ops
    c = gc.RGBD
    inp = gc.rgbd_inputs()
    g = np.load(gc.golden_path("samples"))
    sel = list(range(c["n_frames"])) if tag == "all" else c["selected"]
    n_rays = c["n_rays"] * len(sel)
    n1 = g[f"rgbd_{tag}_u"].shape[0]
    u = torch.zeros(n_rays, c["n_strat"])
    u[:n1] = T(g[f"rgbd_{tag}_u"])
    gg = torch.zeros(n_rays, c["n_surf"] - 1)
    gg[:n1] = T(g[f"rgbd_{tag}_g"])
    out = ops.sample_rays(T(inp["depth"])[sel].to(DEV), T(inp["T_WC"])[sel].to(DEV), T(inp["R"])[sel].to(DEV),
                          T(inp["t"])[sel].to(DEV), (c["fx"], c["fy"], c["cx"], c["cy"]),
                          T(g[f"rgbd_{tag}_pix_h"]).to(DEV), T(g[f"rgbd_{tag}_pix_w"]).to(DEV), u.to(DEV), gg.to(DEV),
                          normals=T(inp["normals"])[sel].to(DEV), frame_ids=torch.tensor(sel),
                          rays_per_frame=c["n_rays"], min_depth=c["min_depth"],
                          dist_behind_surf=c["dist_behind_surf"], trunc_dist=c["trunc_dist"], n_strat=c["n_strat"],
                          n_surf=c["n_surf"])
    n = g[f"rgbd_{tag}_sdf"].shape[0]
    assert out.rows() == n and int(out.counts[0]) == n1
    aux = out.aux.cpu()
    torch.testing.assert_close(out.coords_frame.cpu()[:n], T(g[f"rgbd_{tag}_coords"]), rtol=0, atol=4e-6)
    torch.testing.assert_close(aux[:n, 0:1], T(g[f"rgbd_{tag}_sdf"]), rtol=1e-6, atol=1e-6)
    assert torch.equal(out.sample_frame_ids.cpu()[:n, None], T(g[f"rgbd_{tag}_ids"]))
    edge = ((T(g[f"rgbd_{tag}_sdf"]).abs() - c["trunc_dist"]).abs() < 1e-6)[:, 0]
    assert torch.equal((aux[:n, 1] > 0)[~edge], T(g[f"rgbd_{tag}_valid"])[:, 0][~edge])
    assert torch.equal(aux[:n, 2][~edge], T(g[f"rgbd_{tag}_signs"])[:, 0][~edge])


@pytest.mark.parametrize("B,H,W,rays,n_strat,n_surf", [
    (6, 48, 64, 200, 19, 8),      # ScanNet knobs (configs/rgbd/scannet.yaml:107-111)
    (3, 20, 30, 37, 3, 4),        # PosedSdfRgbd defaults, ragged ray count
    (2, 16, 16, 300, 5, 1),       # surface sample only
    (2, 16, 16, 64, 7, 0),        # no surface samples
    (5, 32, 40, 1000, 64, 2),     # widest stratification the kernel takes
    (1, 8, 8, 1, 2, 3),           # a single ray
])
def test_sample_rays_vs_oracle(B, H, W, rays, n_strat, n_surf):
    depth, normals, Rm, t, Tm = random_frames(B * 1000 + rays, B, H, W)
    g = torch.Generator().manual_seed(rays + n_strat)
    n = B * rays
    ph = torch.randint(0, H, (n,), generator=g)
    pw = torch.randint(0, W, (n,), generator=g)
    pb = torch.arange(B).repeat_interleave(rays)
    u = torch.rand(n, n_strat, generator=g)
    gg = torch.randn(n, max(n_surf - 1, 0), generator=g) * 0.1
    knobs = dict(min_depth=0.07, dist_behind_surf=0.1, trunc_dist=0.15, n_strat=n_strat, n_surf=n_surf)
    intr = (40.0, 38.0, W / 2 - 0.5, H / 2 - 0.5)
    fid = torch.arange(B) * 3 + 1
    res = run_both(depth, normals, Rm, t, Tm, intr, pb, ph, pw, u, gg if n_surf > 1 else None, knobs, frame_ids=fid,
                   rays_per_frame=rays, explicit_b=(rays % 2 == 0))
    compare(*res, knobs, float(t.abs().max()))


def test_sample_rays_shuffled_frames_and_no_normals():
    """pix_b in arbitrary order (not repeat_interleave) and no normal filter."""
    B, H, W, n = 4, 24, 24, 3000
    depth, _, Rm, t, Tm = random_frames(5, B, H, W, nan_depth=6)
    g = torch.Generator().manual_seed(9)
    pb = torch.randint(0, B, (n,), generator=g)
    ph = torch.randint(0, H, (n,), generator=g)
    pw = torch.randint(0, W, (n,), generator=g)
    u = torch.rand(n, 6, generator=g)
    gg = torch.randn(n, 2, generator=g) * 0.1
    knobs = dict(min_depth=0.07, dist_behind_surf=0.2, trunc_dist=0.3, n_strat=6, n_surf=3)
    res = run_both(depth, None, Rm, t, Tm, (20.0, 20.0, 11.5, 11.5), pb, ph, pw, u, gg, knobs)
    compare(*res, knobs, float(t.abs().max()))


def test_sample_rays_nothing_survives_and_empty():
    from miso_amd import ops
    B, H, W, n = 2, 8, 8, 700
    depth = torch.zeros(B, H, W)
    _, _, Rm, t, Tm = random_frames(1, B, H, W)
    z = torch.zeros(n, dtype=torch.int64)
    kn = dict(min_depth=0.07, dist_behind_surf=0.1, trunc_dist=0.15, n_strat=3, n_surf=2, rays_per_frame=n // B)
    out = ops.sample_rays(depth.to(DEV), Tm.to(DEV), Rm.to(DEV), t.to(DEV), (10, 10, 4, 4), z.to(DEV), z.to(DEV),
                          torch.rand(n, 3).to(DEV), torch.randn(n, 1).to(DEV), **kn)
    assert out.counts.cpu().tolist() == [0, 0, 0, 0] and out.rows() == 0
    assert out.aux.abs().sum().item() == 0.0 and out.coords_frame.abs().sum().item() == 0.0
    e = torch.zeros(0, dtype=torch.int64, device=DEV)
    out = ops.sample_rays(depth.to(DEV), Tm.to(DEV), Rm.to(DEV), t.to(DEV), (10, 10, 4, 4), e, e,
                          torch.zeros(0, 3, device=DEV), torch.zeros(0, 1, device=DEV), **kn)
    assert out.rows() == 0


def test_sample_rays_bad_arguments():
    from miso_amd import ops, _lib
    B, H, W, n = 1, 4, 4, 8
    depth, _, Rm, t, Tm = random_frames(2, B, H, W)
    z = torch.zeros(n, dtype=torch.int64, device=DEV)
    kn = dict(min_depth=0.07, dist_behind_surf=0.1, trunc_dist=0.15, n_surf=2)
    with pytest.raises(RuntimeError, match="2002"):    # more bins than MISO_RAY_MAX_BINS
        ops.sample_rays(depth.to(DEV), Tm.to(DEV), Rm.to(DEV), t.to(DEV), (10, 10, 2, 2), z, z,
                        torch.rand(n, 65, device=DEV), torch.randn(n, 1, device=DEV), n_strat=65, rays_per_frame=n, **kn)
    with pytest.raises(RuntimeError, match="2001"):    # no frame assignment at all
        ops.sample_rays(depth.to(DEV), Tm.to(DEV), Rm.to(DEV), t.to(DEV), (10, 10, 2, 2), z, z,
                        torch.rand(n, 3, device=DEV), torch.randn(n, 1, device=DEV), n_strat=3, rays_per_frame=0, **kn)
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.sample_rays(depth, Tm, Rm, t, (10, 10, 2, 2), z, z, torch.rand(n, 3), torch.randn(n, 1), n_strat=3,
                        rays_per_frame=n, **kn)
