"""Submaps that CAN be aligned: every level of every submap samples ONE analytic feature field of the world.

The golden atlas of the tests and bench.py's cfg-4 atlas carry random features -- good for pinning gradients and
trajectories against the reference, useless for asking whether alignment aligns (there is nothing to align to).  Here
a submap's level-l feature at vertex v (submap frame) is F_l(R_true v + t_true): a band-limited sum of sinusoids of the
WORLD position, with wavelengths of several cells of that level, under an envelope that confines it to a SCENE -- a
ball around the world origin that every submap contains with a margin, as every submap of a real run contains the
surfaces it observed and has zero features elsewhere.  Two submaps at their true poses then see the same features up
to trilinear interpolation error, and a perturbed pose has a basin to fall back into.  (Without the envelope the
field runs into the submap bounds, where the zeros-padded trilinear sample of the destination is attenuated within
half a cell of the boundary: a residual that does not vanish at the true pose and pulls a chain of four submaps
0.5 deg / 5 cm away from it.  Real submaps have nothing to compare there.)  Used by
tests/test_align_convergence.py (HIP and the CPU oracle loop) and by tools/demo_synthetic.py's alignment check
(the call sequence of demo/align_submaps.py:240-317: perturb, Fuser.align, trajectory error before / after).
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


LAM_LO, LAM_HI = 5.0, 12.0      # wavelengths in cells of the level


SCENE_RADIUS = 1.2                 # metres: the field vanishes (smoothly) outside this ball around the world origin


def analytic_field(x_world: torch.Tensor, level: int, fdim: int, cell: float, seed: int = 7, amp: float = 0.1,
                   n_waves: int = 4, center=(0.0, 0.0, 0.0), radius: float = SCENE_RADIUS) -> torch.Tensor:
    """(N,3) world positions -> (N, fdim): channel c of level l is sum_j a_j sin(k_j . x + phi_j) with wavelengths of
    LAM_LO .. LAM_HI cells of that level, times the envelope (1 - |x - center|^2 / r^2)^2 inside the scene ball, 0 outside."""
    rs = np.random.RandomState(seed + 101 * level)
    out = torch.zeros((x_world.shape[0], fdim), dtype=torch.float64)
    x = x_world.detach().double().cpu()
    for c in range(fdim):
        for _ in range(n_waves):
            d = rs.standard_normal(3)
            d /= np.linalg.norm(d)
            lam = cell * rs.uniform(LAM_LO, LAM_HI)
            k = torch.tensor(d * (2.0 * math.pi / lam))
            out[:, c] += (amp / n_waves) * rs.uniform(0.5, 1.0) * torch.sin(x @ k + rs.uniform(0, 2 * math.pi))
    xc = x - torch.tensor(center, dtype=torch.float64)
    env = torch.clamp(1.0 - (xc * xc).sum(1) / radius ** 2, min=0.0) ** 2
    return (out * env.unsqueeze(1)).to(torch.float32)


def fill_from_field(atlas, true_poses, seed: int = 7, amp: float = 0.1, center=(0.0, 0.0, 0.0), radius: float = SCENE_RADIUS):
    """Overwrites the feature grids of every submap with samples of the shared field at the submaps' TRUE poses
    (true_poses: list of (R (3,3), t (3,1)) world <- submap)."""
    for s, (R, t) in enumerate(true_poses):
        net = atlas.get_submap(s)
        for l, grid in enumerate(net.features):
            v = grid.vertex_positions().detach().cpu()                      # (Z*Y*X, 3), z-major, submap frame
            w = v.double() @ R.double().cpu().T + t.double().cpu().reshape(1, 3)
            f = analytic_field(w, l, grid.fdim, float(net.cell_sizes[l]), seed=seed, amp=amp, center=center, radius=radius)
            _, C, Z, Y, X = grid.feature.shape
            with torch.no_grad():
                grid.feature.copy_(f.T.reshape(1, C, Z, Y, X).to(grid.feature.device))


def build_atlas(dev, n_submaps: int, bound=((-2.0, 2.0), (-2.0, 2.0), (-2.0, 2.0)), base_cell: float = 0.5,
                scale: int = 4, n_levels: int = 2, fdim: int = 4, hidden: int = 64, spread: float = 0.4, seed: int = 7):
    """n_submaps submaps around the scene: true translations within +-spread metres of the origin, true rotations up
    to ~0.15 rad, every bound containing the scene ball with a margin; features from the shared field.
    Returns (atlas, true_poses)."""
    from miso_amd.grid_opt.models.grid_atlas import GridAtlas
    cfg = {"name": "grid_net", "spatial_dim": 3,
           "decoder": {"type": "mlp", "hidden_dim": hidden, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                       "fix": True, "pretrained_model": None},
           "grid": {"type": "regular", "feature_dim": fdim, "init_stddev": 0.0, "bound": [list(b) for b in bound],
                    "base_cell_size": base_cell, "per_level_scale": scale, "n_levels": n_levels},
           "pose": {"optimize": False, "num_poses": 1}}
    torch.manual_seed(seed)
    atlas = GridAtlas(cfg, device=dev)
    rs = np.random.RandomState(seed)
    true_poses = []
    lb = torch.tensor([list(b) for b in bound], dtype=torch.float32)
    for s in range(n_submaps):
        rv = rs.uniform(-0.15, 0.15, 3) if s else np.zeros(3)
        R = torch.tensor(_rodrigues(rv), dtype=torch.float32)
        t = torch.tensor(rs.uniform(-spread, spread, (3, 1)) if s else np.zeros((3, 1)), dtype=torch.float32)
        atlas.add_submap(lb, R, t, num_poses=1)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
        true_poses.append((R.clone(), t.clone()))      # (add_submap keeps the tensors it is given: not ours to share)
    atlas.to(dev)
    fill_from_field(atlas, true_poses, seed=seed)
    return atlas, true_poses


def _rodrigues(rv):
    rv = np.asarray(rv, dtype=np.float64)
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def perturb(atlas, true_poses, deg: float = 5.0, metres: float = 0.3, seed: int = 55):
    """Submaps 1.. get a base pose off by exactly `deg` degrees (random axis) and `metres` (random direction), as
    demo/align_submaps.py:267-273 does with Gaussian noise; submap 0 keeps its true pose (it is the fixed one)."""
    rs = np.random.RandomState(seed)
    for s in range(1, atlas.num_submaps):
        axis = rs.standard_normal(3)
        axis /= np.linalg.norm(axis)
        d = rs.standard_normal(3)
        d /= np.linalg.norm(d)
        R, t = true_poses[s]
        Rn = torch.tensor(_rodrigues(axis * math.radians(deg)), dtype=torch.float32)
        atlas.set_submap_pose(s, R @ Rn, t + torch.tensor(d * metres, dtype=torch.float32).reshape(3, 1))


def pose_errors(atlas, true_poses):
    """(max rotation error in degrees, max translation error in metres) of submaps 1.. against the truth."""
    worst_deg = worst_m = 0.0
    for s in range(1, atlas.num_submaps):
        R, t = (v.detach().cpu().double() for v in atlas.updated_submap_pose(s))
        Rt, tt = (v.double() for v in true_poses[s])
        c = float(((R @ Rt.T).trace() - 1.0) / 2.0)
        worst_deg = max(worst_deg, math.degrees(math.acos(max(-1.0, min(1.0, c)))))
        worst_m = max(worst_m, float((t - tt).norm()))
    return worst_deg, worst_m


ALIGN_CFG = {"level_iters": 100, "finetune_iters": 100, "learning_rate": 0.01, "loss_type": "L2",
             "stability_thresh": 0.0, "subsample_points": None, "latent_levels": [0, 1], "skip_finetune": True,
             "pose_reg_weight": 0.0, "verbose": True, "save_iterations": True}      # configs/rgbd/scannet.yaml:55-66


class OneItem(torch.utils.data.Dataset):
    def __len__(self):
        return 1

    def __getitem__(self, i):
        return 0


def run(dev, n_submaps=2, deg=5.0, metres=0.3, align_cfg=None):
    """perturb -> Fuser.align -> errors before / after; returns (before, after, info, atlas)."""
    from miso_amd.grid_opt.slam.fuser import Fuser
    atlas, true_poses = build_atlas(dev, n_submaps)
    perturb(atlas, true_poses, deg, metres)
    before = pose_errors(atlas, true_poses)
    cfg = {"align": dict(align_cfg or ALIGN_CFG), "device": str(dev)}
    info = Fuser(model=atlas, dataset=OneItem(), cfg=cfg).align()
    return before, pose_errors(atlas, true_poses), info, atlas


if __name__ == "__main__":
    dev = sys.argv[1] if len(sys.argv) > 1 else "cuda:0"
    for n in (2, 4):
        b, a, info, _ = run(dev, n)
        print(f"{n} submaps: pose error {b[0]:.2f} deg / {b[1]:.3f} m  ->  {a[0]:.3f} deg / {a[1]:.4f} m   "
              f"(gpu_time {info['gpu_time_sec']:.3f} s)")
