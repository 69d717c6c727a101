"""Shared, seed-pinned input definitions for the golden vectors.

Used by ``tools/make_goldens.py`` (which imports the reference in the build
container and writes ``tests/golden/*.npz``) and by the tests (which rebuild
the same inputs and compare oracle / HIP outputs with the stored reference
outputs).  Inputs come from ``np.random.RandomState`` (a stream that numpy
guarantees stable across versions), so large grids need not be committed.
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name + ".npz")


def grid_shape(bound, cell, fdim):
    """(1,C,Z,Y,X) exactly as grid_opt/models/grid_modules.py:47-57."""
    b = np.asarray(bound, dtype=np.float32)
    length = (b[:, 1] - b[:, 0])
    size = np.ceil(length / cell).astype(int)
    return (1, fdim, int(size[2]), int(size[1]), int(size[0]))


def model_cfg(bound, base_cell, scale, n_levels, fdim, hidden, hidden_layers=1,
              num_poses=1, optimize_pose=False, init_stddev=0.0, decoder_fix=True,
              second_order=False):
    """cfg['model'] dict with the keys read at grid_net.py:35-43,106-111,140-141."""
    cfg = {
        "name": "grid_net",
        "spatial_dim": 3,
        "decoder": {"type": "mlp", "hidden_dim": hidden, "hidden_layers": hidden_layers,
                    "out_dim": 1, "pos_invariant": True, "fix": decoder_fix,
                    "pretrained_model": None},
        "grid": {"type": "regular", "feature_dim": fdim, "init_stddev": init_stddev,
                 "bound": [list(map(float, r)) for r in bound],
                 "base_cell_size": base_cell, "per_level_scale": scale,
                 "n_levels": n_levels},
        "pose": {"optimize": optimize_pose, "num_poses": num_poses},
    }
    if second_order:
        cfg["grid"]["second_order_grid_sample"] = True
    return cfg


CASES = {
    # small, non-cubic, two levels -- everything stored
    "small": dict(bound=[[-1.0, 1.3], [-0.7, 0.9], [0.0, 2.1]], base_cell=0.4, scale=2,
                  n_levels=2, fdim=4, hidden=32, n_points=512, seed=11),
    # BASELINE cfg-1: 1 level 64^3, C=4, MLP 4-32-32-1, 4096 pts
    "cfg1": dict(bound=[[-1.0, 1.0]] * 3, base_cell=2.0 / 64, scale=2, n_levels=1,
                 fdim=4, hidden=32, n_points=4096, seed=1),
    # BASELINE cfg-2 shapes: 3 levels 32/64/128^3, C=8, MLP 24-64-64-1 (8192-pt subset)
    "cfg2": dict(bound=[[-1.0, 1.0]] * 3, base_cell=2.0 / 32, scale=2, n_levels=3,
                 fdim=8, hidden=64, n_points=8192, seed=2),
}


def level_cells(case):
    return [case["base_cell"] / (case["scale"] ** l) for l in range(case["n_levels"])]


def make_features(case, std=1e-2):
    """Per-level feature grids (1,C,Z,Y,X) fp32, N(0, std^2)."""
    rs = np.random.RandomState(case["seed"])
    feats = []
    for cell in level_cells(case):
        shp = grid_shape(case["bound"], cell, case["fdim"])
        feats.append((rs.standard_normal(shp) * std).astype(np.float32))
    return feats


def make_stability(case):
    rs = np.random.RandomState(case["seed"] + 1000)
    out = []
    for cell in level_cells(case):
        shp = grid_shape(case["bound"], cell, 1)
        out.append((1.0 + 0.1 * rs.standard_normal(shp)).astype(np.float32))
    return out


def make_decoder(case, in_dim=None, hidden_layers=1, out_dim=1):
    """MLPNet state-dict arrays (keys network.{0,2,4}.{weight,bias})."""
    rs = np.random.RandomState(case["seed"] + 2000)
    f = case["fdim"] * case["n_levels"] if in_dim is None else in_dim
    h = case["hidden"]
    dims = [f] + [h] * (hidden_layers + 1) + [out_dim]
    sd = {}
    for i in range(len(dims) - 1):
        k = 1.0 / np.sqrt(dims[i])
        sd[f"network.{2 * i}.weight"] = rs.uniform(-k, k, (dims[i + 1], dims[i])).astype(np.float32)
        sd[f"network.{2 * i}.bias"] = rs.uniform(-k, k, (dims[i + 1],)).astype(np.float32)
    return sd


def make_points(case, n_edge=64):
    """Uniform-in-bbox points + edge points (on the bound, +-half a finest cell
    from it, and outside)."""
    rs = np.random.RandomState(case["seed"] + 3000)
    b = np.asarray(case["bound"], dtype=np.float32)
    n = case["n_points"]
    pts = rs.uniform(0, 1, (n, 3)).astype(np.float32) * (b[:, 1] - b[:, 0]) + b[:, 0]
    fine = level_cells(case)[-1]
    edge = rs.uniform(0, 1, (n_edge, 3)).astype(np.float32) * (b[:, 1] - b[:, 0]) + b[:, 0]
    for i in range(n_edge):
        ax = i % 3
        side = (i // 3) % 2
        kind = (i // 6) % 5
        base = b[ax, side]
        sgn = 1.0 if side == 1 else -1.0
        off = [0.0, -0.5 * fine * sgn, 0.5 * fine * sgn, 0.25 * fine * sgn, 3.0 * fine * sgn][kind]
        edge[i, ax] = base + off
    return np.concatenate([pts, edge.astype(np.float32)], 0)


def make_targets(case, n):
    rs = np.random.RandomState(case["seed"] + 4000)
    sdf = (0.1 * rs.standard_normal((n, 1))).astype(np.float32)
    valid = (rs.uniform(0, 1, (n, 1)) > 0.1).astype(np.float32)
    sign = (rs.uniform(0, 1, (n, 1)) > 0.5).astype(np.float32)
    weight = rs.uniform(0.5, 1.5, (n, 1)).astype(np.float32)
    return sdf, valid, sign, weight


def sample_indices(numel, k=256, seed=7):
    rs = np.random.RandomState(seed)
    return rs.randint(0, numel, size=k).astype(np.int64)


# --- atlas / alignment case -------------------------------------------------
ATLAS = dict(bound=[[-2.0, 2.0], [-1.0, 1.0], [-2.0, 2.0]], base_cell=0.5, scale=5,
             n_levels=2, fdim=4, hidden=64, seed=21, n_submaps=3, n_points=1024)


def rodrigues(rotvec):
    rotvec = np.asarray(rotvec, dtype=np.float64)
    th = np.linalg.norm(rotvec)
    if th < 1e-12:
        return np.eye(3)
    k = rotvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def atlas_inputs():
    """Per-submap features (smooth + noise so that gradients are informative),
    world poses on a line with ~50 % overlap, and small perturbations."""
    c = ATLAS
    rs = np.random.RandomState(c["seed"])
    subs = []
    for s in range(c["n_submaps"]):
        feats = []
        for cell in level_cells(c):
            shp = grid_shape(c["bound"], cell, c["fdim"])
            f = (rs.standard_normal(shp) * 1e-1).astype(np.float32)
            # leave a slab of exact zeros so the norm-threshold pruning is exercised
            f[..., : max(1, shp[-1] // 8)] = 0.0
            feats.append(f)
        rot = rodrigues(rs.uniform(-0.15, 0.15, 3)).astype(np.float32)
        t = np.array([[1.8 * s], [0.1 * s], [-0.2 * s]], dtype=np.float32) \
            + rs.uniform(-0.05, 0.05, (3, 1)).astype(np.float32)
        dr = rs.uniform(-0.05, 0.05, (1, 3)).astype(np.float32)
        dt = rs.uniform(-0.1, 0.1, (3, 1)).astype(np.float32)
        subs.append(dict(features=feats, R=rot, t=t, dr=dr, dt=dt))
    return subs


def atlas_stability(s):
    """Per-level stability grids (1,1,Z,Y,X) of submap s, values in [0,1): what GridNet.feature_stability holds once a
    stability estimate has been written (pairwise_loss_latent's stability_thresh branch, align/miso.py:160-171)."""
    c = ATLAS
    rs = np.random.RandomState(c["seed"] + 300 + s)
    return [rs.uniform(0.0, 1.0, grid_shape(c["bound"], cell, 1)).astype(np.float32) for cell in level_cells(c)]


ATLAS_BRANCHES = dict(pair=(0, 1), stability_thresh=0.4, trunc_factor=1.5, subsample_points=300, subsample_seed=4242)


def atlas_world_points():
    c = ATLAS
    rs = np.random.RandomState(c["seed"] + 5)
    lo = np.array([-2.5, -1.5, -2.8], dtype=np.float32)
    hi = np.array([6.0, 1.5, 2.5], dtype=np.float32)
    return (rs.uniform(0, 1, (c["n_points"], 3)).astype(np.float32) * (hi - lo) + lo)


# --- sample generation (depth frames / lidar frames -> SDF rows) --------------
RGBD = dict(seed=31, n_frames=4, H=24, W=32, fx=30.0, fy=28.0, cx=15.5, cy=11.5, n_rays=48, n_strat=5, n_surf=4,
            min_depth=0.07, dist_behind_surf=0.1, trunc_dist=0.15, selected=[2, 0, 3])


def rgbd_inputs(case=None):
    """Depth frames with holes (0 = no return), one NaN depth, normals with NaNs, random keyframe poses."""
    c = case or RGBD
    rs = np.random.RandomState(c["seed"])
    B, H, W = c["n_frames"], c["H"], c["W"]
    depth = rs.uniform(0.4, 4.0, (B, H, W)).astype(np.float32)
    depth[rs.uniform(0, 1, (B, H, W)) < 0.2] = 0.0
    depth[min(2, B - 1), 3, 5] = np.nan
    normals = rs.standard_normal((B, H, W, 3)).astype(np.float32)
    normals[rs.uniform(0, 1, (B, H, W)) < 0.1] = np.nan
    R = np.stack([rodrigues(rs.uniform(-0.8, 0.8, 3)) for _ in range(B)]).astype(np.float32)
    t = rs.uniform(-3.0, 3.0, (B, 3, 1)).astype(np.float32)
    T = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    T[:, :3, :3] = R
    T[:, :3, 3:] = t
    return dict(depth=depth, normals=normals, R=R, t=t, T_WC=T)


LIDAR = dict(seed=41, n_frames=3, n_points=[96, 64, 80], frame_samples=72, frame_batchsize=200, near_surface_n=4,
             near_surface_std=0.1, free_space_n=2, behind_surface_n=1, trunc_dist=0.5, min_dist_ratio=0.3,
             max_range=60.0)


def lidar_inputs(case=None):
    """World-frame surface points per frame (ranges 2..30 m around the sensor) and sensor poses."""
    c = case or LIDAR
    rs = np.random.RandomState(c["seed"])
    frames = []
    for f in range(c["n_frames"]):
        R = rodrigues(rs.uniform(-0.5, 0.5, 3)).astype(np.float32)
        t = rs.uniform(-5.0, 5.0, (3, 1)).astype(np.float32)
        d = rs.standard_normal((c["n_points"][f], 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = t.reshape(1, 3).astype(np.float64) + d * rs.uniform(2.0, 30.0, (c["n_points"][f], 1))
        frames.append(dict(R=R, t=t, points_global=pts))
    return frames


# --- extra rows: SDF-space pair loss, scatter-average pooling ------------------
def atlas_second_kf_pose(s):
    """Pose (in its submap) of the second keyframe of submap s; the anchor keyframe sits at identity."""
    rs = np.random.RandomState(ATLAS["seed"] + 50 + s)
    R = rodrigues(rs.uniform(-0.3, 0.3, 3)).astype(np.float32)
    t = rs.uniform(-0.4, 0.4, (3, 1)).astype(np.float32)
    return R, t


def atlas_sdf_batch(n=900):
    """One dataset batch in keyframe frames: rows spread over the 2 keyframes of each of the 3 submaps (global ids
    0..5), some rows invalid, some far enough out to leave the other submap's bound."""
    rs = np.random.RandomState(ATLAS["seed"] + 77)
    b = np.asarray(ATLAS["bound"], dtype=np.float32)
    coords = (rs.uniform(-0.1, 1.1, (n, 3)).astype(np.float32) * (b[:, 1] - b[:, 0]) + b[:, 0])
    ids = rs.randint(0, 2 * ATLAS["n_submaps"], size=(n, 1)).astype(np.int64)
    valid = (rs.uniform(0, 1, (n, 1)) > 0.15).astype(np.float32)
    sdf = (0.1 * rs.standard_normal((n, 1))).astype(np.float32)
    return dict(coords_frame=coords[None], sample_frame_ids=ids[None], weights=np.ones((1, n, 1), np.float32)), \
        dict(sdf=sdf[None], sdf_valid=valid[None], sdf_signs=np.zeros((1, n, 1), np.float32))


POOL = dict(seed=61, bound=[[-1.0, 1.0], [-0.5, 0.55], [0.0, 1.3]], cell=0.3, n=700, d=5)


def pool_inputs():
    """Points in and slightly around the bound (outside ones are clamped into the border cells) + features."""
    c = POOL
    rs = np.random.RandomState(c["seed"])
    b = np.asarray(c["bound"], dtype=np.float32)
    pts = rs.uniform(-0.08, 1.08, (c["n"], 3)).astype(np.float32) * (b[:, 1] - b[:, 0]) + b[:, 0]
    pts[:40] = pts[40:80] + 1e-3 * rs.standard_normal((40, 3)).astype(np.float32)      # crowded cells
    feats = rs.standard_normal((c["n"], c["d"])).astype(np.float32)
    return pts, feats


def local_opt_cfg(loss_name):
    """cfg dict for grid_opt.local_opt (cfg['loss'] -> cfg_loss, cfg['train'] -> GridTrainer)."""
    return {"device": "cpu",
            "model": {"name": "grid_net"},
            "loss": {"name": loss_name, "trunc_weight": 5.0, "trunc_distance": 0.15, "noise_std": 0, "orien_loss": 0,
                     "eik_weight": 0, "grad_weight": 0, "eik_apply_dist": 0.1, "smooth_weight": 0, "smooth_std": 0.05,
                     "loss_type": "L1", "slam_mode": False, "pose_reg_weight": 0, "pose_thresh_m": 1.0,
                     "pose_thresh_rad": 1.0, "feat_reg_weight": 0},
            "train": {"trainer": "grid", "verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1,
                      "ckpt_every": -1, "eval_every": -1, "eval_metric": None, "pretrained_model": None,
                      "log_dir": "/tmp/miso_golden_log", "relchange_tol": 0, "max_epochs_in_level": 2,
                      "grid_training_mode": "joint"}}


# --- rigid-map / point-cloud helpers of utils_geometry ---------------------------
def geometry_inputs():
    rs = np.random.RandomState(91)
    F = 4
    R = np.stack([rodrigues(rs.uniform(-1.0, 1.0, 3)) for _ in range(F)]).astype(np.float32)
    t = rs.uniform(-2.0, 2.0, (F, 3, 1)).astype(np.float32)
    dr = rs.uniform(-0.2, 0.2, (F, 3)).astype(np.float32)
    dt = rs.uniform(-0.3, 0.3, (F, 3, 1)).astype(np.float32)
    spans = np.array([[0, 50], [50, 50], [50, 130], [130, 200]], dtype=np.int64)     # one empty frame
    pts = rs.uniform(-3.0, 3.0, (200, 3)).astype(np.float32)
    cloud = (rs.standard_normal((3000, 3)) * np.array([6.0, 6.0, 1.5])).astype(np.float32)
    stamps = rs.uniform(0, 1, (3000, 1)).astype(np.float32)
    return dict(R=R, t=t, dr=dr, dt=dt, spans=spans, pts=pts, cloud=cloud, stamps=stamps)
