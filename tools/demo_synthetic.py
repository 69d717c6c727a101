"""The two MISO demos end to end on synthetic RGB-D frames, through the reference's import names.

demo/build_submaps.py (main_scannet, :125-141) and demo/align_submaps.py (main_scannet, :240-317) need a ScanNet
scene on disk, the pretrained decoder_indoor.pt and Open3D windows.  This script executes the SAME call sequence --
same modules, classes and functions, imported as ``grid_opt.*`` via ``miso_amd.compat`` -- with the three things that
cannot exist offline replaced: the dataset is ``PosedSdfRgbd.from_frames`` on depth images rendered from an analytic
room instead of ``utils_scannet.create_scannet_dataset`` (:48-50), the decoder keeps its seeded random weights
(frozen, as configs/rgbd/scannet.yaml:16), and ``o3d.visualization.draw_geometries`` calls are dropped.

  build:  dry-run System (tracking / mapping disabled: creates submaps + keyframes)  -> per submap Mapper.mapping
          -> utils_sdf.save_mesh (coarse level, then both)  -> torch.save(grid_atlas)
  align:  torch.load(grid_atlas)  -> perturb submap poses (10 deg / 0.5 m, seed 55)  -> Fuser.align (hierarchical
          latent alignment, verbose + save_iterations as the reference config)  -> trajectory error before / after
          (utils_eval.evo_trajectory_error)  -> alignment_result.json
  align, shared field:  the same atlas file and the same call sequence once more, with the feature grids replaced by
          samples of ONE analytic field of the world at the true submap poses (tools/shared_field.py).  Upstream the
          pretrained decoder is what makes two submaps describe the same surface with the same latent features; with
          the random frozen decoder of this offline demo the mapped features of two submaps need not agree, so the
          first alignment only has to run.  On the shared field there IS a pose to come back to: the trajectory error
          must fall by 5x (asserted).

    python tools/demo_synthetic.py --save_dir /tmp/miso_demo [--quick]
"""
import argparse
import json
import os
import sys
from math import radians
from os.path import join

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import miso_amd.compat  # noqa: E402,F401  (grid_opt.* and cuda_gridsample resolve to this package)

from grid_opt.configs import *  # noqa: E402,F401,F403
from grid_opt.utils.utils_sdf import *  # noqa: E402,F401,F403
from grid_opt.slam.fuser import Fuser  # noqa: E402
from grid_opt.slam.mapper import Mapper  # noqa: E402
from grid_opt.slam.system import System  # noqa: E402
from grid_opt.datasets.sdf_rgbd import PosedSdfRgbd  # noqa: E402
from grid_opt.models.grid_atlas import GridAtlas  # noqa: E402
from grid_opt.utils.utils_data import CameraParameters  # noqa: E402
import grid_opt.utils.utils_eval as utils_eval  # noqa: E402
import grid_opt.utils.utils_geometry as utils_geometry  # noqa: E402
import grid_opt.utils.utils_sdf as utils_sdf  # noqa: E402
import grid_opt.utils.utils as utils  # noqa: E402
import torch  # noqa: E402

ROOM = np.array([[0.0, 8.0], [0.0, 6.0], [0.0, 3.0]])       # an axis-aligned room, metres


def render_depth(R, t, cam):
    """z-depth of the room's walls seen by a pin-hole camera at (R, t) (camera -> world, +z forward)."""
    v, u = np.meshgrid(np.arange(cam.H), np.arange(cam.W), indexing="ij")
    d_c = np.stack(((u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u, dtype=np.float64)), -1)
    d_w = d_c @ R.T
    with np.errstate(divide="ignore", invalid="ignore"):
        hit = np.where(d_w > 0, (ROOM[:, 1] - t) / d_w, (ROOM[:, 0] - t) / d_w)
    return hit.min(axis=-1).astype(np.float32)               # ray parameter = z-depth because d_c.z = 1


def synthetic_sequence(n_kf, cam):
    """A camera walking through the room, turning as it goes: (n_kf, H, W) depth, (n_kf,3,3), (n_kf,3,1) poses."""
    depth, Rs, ts = [], [], []
    for k in range(n_kf):
        s = k / max(n_kf - 1, 1)
        yaw = radians(-40.0 + 170.0 * s)
        pitch = radians(8.0 * np.sin(3.0 * s))
        # camera axes in the world: z forward (horizontal), y down
        fwd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), -np.sin(pitch)])
        down = np.array([0.0, 0.0, -1.0])
        right = np.cross(down, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack((right, down, fwd), axis=1)
        t = np.array([2.0 + 4.0 * s, 2.0 + 2.0 * np.sin(2.5 * s), 1.4 + 0.2 * np.cos(4.0 * s)])
        depth.append(render_depth(R, t, cam))
        Rs.append(R)
        ts.append(t.reshape(3, 1))
    f32 = lambda a: torch.tensor(np.stack(a), dtype=torch.float32)       # noqa: E731
    return f32(depth), f32(Rs), f32(ts)


def create_configs(args, dataset):
    """The keys of configs/base.yaml + configs/rgbd/scannet.yaml the two demos end up using, with the overrides of
    create_configs_scannet (demo/build_submaps.py:26-43, demo/align_submaps.py:38-60); grids sized for the room."""
    cfg = {
        "device": args.device,
        "model": {"name": "grid_net", "spatial_dim": 3,
                  "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                              "fix": True, "pretrained_model": None},                      # scannet.yaml:11-18
                  "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 0.0, "bound": ROOM.tolist(),
                           "base_cell_size": args.base_cell, "per_level_scale": 5, "n_levels": 2,
                           "second_order_grid_sample": True},                            # scannet.yaml:19-26
                  "pose": {"optimize": False, "num_poses": dataset.num_kfs}},
        "tracking": {"solver": "adam", "learning_rate": 1e-3, "loss_type": "L1", "trunc_dist": 0.15, "gm_scale_sdf": 0.1,
                     "lm_lambda": 1e-4, "lm_max_iter": 10, "lm_tol_deg": 0.01, "lm_tol_m": 0.001, "verbose": False,
                     "disable": True},
        # (the reference maps with lr 1e-3 on top of its pretrained decoder; a random frozen decoder needs larger steps)
        "mapping": {"learning_rate": 1e-2, "loss_type": "L1", "weight_sdf": 1.0, "weight_eik": 0.0, "weight_fs": 0.1,
                    "trunc_dist": 0.15, "finite_diff_eps": 0.03, "grad_method": "finitediff", "eik_trunc_dist": 0.024,
                    "verbose": False, "max_replay_frames": 10, "max_replay_freq": 10, "gm_scale_sdf": 0.1,
                    "disable": True},
        "align": {"level_iters": args.align_iters, "finetune_iters": args.align_iters, "learning_rate": 0.01,
                  "loss_type": "L2", "stability_thresh": 0.0, "subsample_points": None, "latent_levels": [0, 1],
                  "skip_finetune": True, "pose_reg_weight": 0.0, "verbose": True, "save_iterations": True},
        "system": {"init_odom": "external", "submap_size": args.submap_size,
                   "submap_local_bound": [[-args.local, args.local], [-args.local, args.local], [-args.local, args.local]],
                   "submap_fov_thresh": 0.0, "save_submap_mesh": False, "log_dir": join(args.save_dir, "system")},
        "train": {"trainer": "base", "verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1,
                  "ckpt_every": -1, "eval_every": -1, "eval_metric": None, "pretrained_model": None,
                  "log_dir": join(args.save_dir, "train"), "relchange_tol": 0, "max_epochs_in_level": 100,
                  "grid_training_mode": "joint"},
        "visualizer": {"enable": False},
    }
    return cfg


def initialize(args, dataset):
    """demo/build_submaps.py:46-73: a dry run over the sequence that only creates the submap / keyframe structure."""
    cfg = create_configs(args, dataset)
    grid_atlas = GridAtlas(cfg['model'], device=cfg['device'], dtype=torch.float32)
    grid_atlas.to(cfg['device'])
    R_world_origin, t_world_origin = dataset.true_kf_pose_in_world(0)
    system = System(model=grid_atlas, dataset_track=dataset, dataset_map=dataset, cfg=cfg,
                    R_world_origin=R_world_origin, t_world_origin=t_world_origin, verbose=False)
    system.run()
    return cfg, grid_atlas


def submap_mapping(cfg, grid_atlas, dataset, submap_id, iterations, level_iterations):
    """demo/build_submaps.py:76-91."""
    cfg['mapping']['verbose'] = False
    cfg['mapping']['disable'] = False
    size = cfg['system']['submap_size']
    frame_start, frame_end = size * submap_id, min(size * (submap_id + 1), dataset.num_kfs)
    mapper = Mapper(model=grid_atlas.get_submap(submap_id), dataset=dataset, cfg=cfg)
    mapper.mapping(mapping_kfs=range(frame_start, frame_end), iterations=iterations, level_iterations=level_iterations)


def save_submap(grid_atlas, submap_id, save_dir, resolution, postfix):
    """demo/build_submaps.py:93-103 without the viewer."""
    submap = grid_atlas.get_submap(submap_id)
    T = utils_geometry.pose_matrix(*grid_atlas.updated_submap_pose(submap_id))
    mesh = utils_sdf.save_mesh(submap, submap.bound, transform=T, resolution=resolution, device=str(submap.device),
                               save_path=join(save_dir, f'submap_{submap_id}_{postfix}.ply'))
    return mesh


def evaluate_alignment_error(grid_atlas, dataset):
    """demo/align_submaps.py:123-146."""
    n = grid_atlas.num_submaps
    R_true, t_true = utils_geometry.identity_rotations(n), torch.zeros((n, 3, 1))
    R_sol, t_sol = utils_geometry.identity_rotations(n), torch.zeros((n, 3, 1))
    for s in range(n):
        anchor = grid_atlas.anchor_kf_for_submap(s)
        R_sol[s], t_sol[s] = (v.detach().cpu() for v in grid_atlas.updated_submap_pose(s))
        R_true[s], t_true[s] = dataset.true_kf_pose_in_world(anchor)
    m_t = utils_eval.evo_trajectory_error(R_true, t_true, R_sol, t_sol, align=True,
                                          pose_relation=utils_eval.PoseRelation.translation_part).get_all_statistics()
    m_R = utils_eval.evo_trajectory_error(R_true, t_true, R_sol, t_sol, align=True,
                                          pose_relation=utils_eval.PoseRelation.rotation_part).get_all_statistics()
    return {'rmse_tran (cm)': 100 * m_t['rmse'], 'rmse_deg': utils_geometry.chordal_to_degree(m_R['rmse'])}


def perturb_and_align(model_path, cfg, dataset, noise_rot, noise_tra, shared_field=False):
    """demo/align_submaps.py:262-317: load, perturb, Fuser.align, trajectory error before / after."""
    grid_atlas = torch.load(model_path, weights_only=False)
    if shared_field:
        import tools.shared_field as SF
        true = [dataset.true_kf_pose_in_world(grid_atlas.anchor_kf_for_submap(s)) for s in range(grid_atlas.num_submaps)]
        centre = tuple(float(v) for v in ROOM.mean(axis=1))
        SF.fill_from_field(grid_atlas, [(R.cpu(), t.cpu()) for R, t in true], center=centre, radius=1.4)
    for i in range(1, grid_atlas.num_submaps):
        R, t = grid_atlas.initial_submap_pose(i)
        grid_atlas.set_submap_pose(i, R @ noise_rot[i].to(R), t + noise_tra[i].reshape(3, 1).to(t))
    metrics_bef = evaluate_alignment_error(grid_atlas, dataset)
    align_info = Fuser(model=grid_atlas, dataset=dataset, cfg=cfg).align()
    metrics_aft = evaluate_alignment_error(grid_atlas, dataset)
    return grid_atlas, align_info, metrics_bef, metrics_aft


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--save_dir', type=str, default='./results/demo/synthetic')
    ap.add_argument('--device', type=str, default='cuda:0')
    ap.add_argument('--quick', action='store_true', help='small images, few iterations (the GPU test)')
    args = ap.parse_args()
    if args.quick:
        H, W, n_kf, args.submap_size, n_rays = 60, 80, 12, 4, 120
        args.map_iters, args.level_iters_map, args.align_iters, args.mesh_res = 120, 40, 100, 48
        args.base_cell, args.local = 0.5, 6.0
    else:
        H, W, n_kf, args.submap_size, n_rays = 120, 160, 30, 10, 200
        args.map_iters, args.level_iters_map, args.align_iters, args.mesh_res = 300, 50, 100, 128
        args.base_cell, args.local = 0.5, 8.0
    np.random.seed(55)
    torch.manual_seed(55)
    utils.cond_mkdir(args.save_dir)
    cam = CameraParameters(fx=0.9 * W, fy=0.9 * W, cx=(W - 1) / 2, cy=(H - 1) / 2, H=H, W=W)
    depth, R_gt, t_gt = synthetic_sequence(n_kf, cam)
    dataset = PosedSdfRgbd.from_frames(depth, R_gt, t_gt, cam, n_rays=n_rays, n_surf_samples=8, n_strat_samples=19,
                                       trunc_dist=0.15, min_depth=0.07, max_depth=12.0, device=args.device)
    # ---------------------------------------------------------------- build_submaps.main_scannet
    model_path = join(args.save_dir, 'grid_atlas.pth')
    cfg, grid_atlas = initialize(args, dataset)
    print(f"dry run: {grid_atlas.num_submaps} submaps, {grid_atlas.num_keyframes} keyframes")
    for i in range(grid_atlas.num_submaps):
        submap_mapping(cfg, grid_atlas, dataset, i, args.map_iters, args.level_iters_map)
        grid_atlas.ignore_level(1)
        save_submap(grid_atlas, i, join(args.save_dir, 'submaps'), args.mesh_res, 'coarse')
        grid_atlas.include_level(1)
        mesh = save_submap(grid_atlas, i, join(args.save_dir, 'submaps'), args.mesh_res, 'fine')
        print(f"submap {i}: mapped, mesh of {len(mesh.triangles)} triangles")
    torch.save(grid_atlas, model_path)
    # ---------------------------------------------------------------- align_submaps.main_scannet
    n_sub = grid_atlas.num_submaps
    noise_rot = utils_geometry.wrapped_gaussian_rotations(n_sub, std_rad=radians(10.0))
    noise_tra = utils_geometry.gaussian_translations(n_sub, stddev=0.50)
    grid_atlas, align_info, metrics_bef, metrics_aft = perturb_and_align(model_path, cfg, dataset, noise_rot, noise_tra)
    print("Before alignment metrics:\n", json.dumps(metrics_bef, indent=4))
    print("After alignment metrics:\n", json.dumps(metrics_aft, indent=4))
    # the same file, the same perturbation, the same calls on features that two submaps agree on
    _, info_sf, sf_bef, sf_aft = perturb_and_align(model_path, cfg, dataset, noise_rot, noise_tra, shared_field=True)
    print("Shared-field features, before:\n", json.dumps(sf_bef, indent=4))
    print("Shared-field features, after:\n", json.dumps(sf_aft, indent=4))
    with open(join(args.save_dir, 'alignment_result.json'), 'w') as f:
        json.dump({'before_alignment': metrics_bef, 'after_alignment': metrics_aft,
                   'shared_field_before': sf_bef, 'shared_field_after': sf_aft}, f, indent=4)
    assert sf_aft['rmse_tran (cm)'] <= 0.2 * sf_bef['rmse_tran (cm)'] and sf_aft['rmse_deg'] <= 0.2 * sf_bef['rmse_deg'], \
        "alignment on the shared field did not bring the submaps back"
    it = align_info['hier_latent_level1_L2']['iteration_results']
    assert sorted(it) == list(range(args.align_iters + 1)) and it[0].shape == (grid_atlas.num_submaps, 4, 4)
    print(f"alignment: {len(it)} pose snapshots per level, gpu_time {align_info['gpu_time_sec']:.3f} s")


if __name__ == "__main__":
    main()
