"""Frame-to-submap tracking: Adam on the keyframe pose, or Gauss-Newton / LM with an
explicit N x 6 pose Jacobian and Geman-McClure weights (reference: grid_opt/slam/tracker.py)."""
import logging
import math
from copy import deepcopy
from typing import List

import torch
from torch.utils.data import DataLoader
from miso_amd.grid_opt.utils.utils import collate_batch_of_one

import miso_amd.grid_opt.utils.utils as utils
import miso_amd.grid_opt.utils.utils_geometry as utils_geometry
from miso_amd.grid_opt.diff import gradient3d
from miso_amd.grid_opt.loss import MisoLossTracking
from miso_amd.grid_opt.models.grid_net import GridNet
from miso_amd.grid_opt.trainer import Trainer
from miso_amd.so3 import hat

logger = logging.getLogger(__name__)


class Tracker:
    def __init__(self, model: GridNet, dataset, cfg: dict):
        assert isinstance(model, GridNet), "Model must be an instance of GridNet."
        self.grid = model
        self.dataset = dataset
        self.train_loader = DataLoader(dataset, shuffle=True, batch_size=1, num_workers=0, collate_fn=collate_batch_of_one)
        self.cfg = cfg
        t = cfg['tracking']
        self.lr = t['learning_rate']
        self.verbose = t['verbose']
        self.gm_scale_sdf = t['gm_scale_sdf']
        self.lm_lambda = t['lm_lambda']
        self.lm_max_iter = t['lm_max_iter']
        self.lm_tol_deg = t['lm_tol_deg']
        self.lm_tol_m = t['lm_tol_m']
        self.loss_type = t['loss_type']
        self.trunc_dist = t['trunc_dist']
        self.solver = t['solver']
        if self.solver == 'adam':
            self.loss_fn = MisoLossTracking(weight_sdf=1.0, loss_type=self.loss_type, trunc_dist=self.trunc_dist,
                                            gm_scale_sdf=self.gm_scale_sdf)
        self.disable = bool(t.get('disable', False))
        self.fused = bool(t.get('fused', True))      # the device-side LM step / Adam window (ours; not a reference key)
        self.latest_fov_overlap = 1.0

    def initialize_window(self, head_kf, tail_kf):
        """Propagate odometry to initialise the keyframe poses in [head_kf, tail_kf)."""
        with torch.no_grad():
            for dst in range(head_kf, tail_kf):
                src = dst - 1
                assert src >= 0
                T_src = utils_geometry.pose_matrix(*self.grid.updated_kf_pose_in_world(src))
                T_dst = T_src @ self.dataset.get_odometry_at_pose(src).to(T_src)
                self.grid.set_initial_kf_pose(dst, T_dst[:3, :3], T_dst[:3, [3]])

    def track_window(self, optimize_kfs: List[int], iterations=10):
        self.grid.lock_feature()
        self.grid.unlock_pose()
        self.grid.lock_all_pose_indices()
        for kf in optimize_kfs:
            self.grid.unlock_pose_index(kf)
        self.dataset.select_keyframes(optimize_kfs)
        cfg_train = deepcopy(self.cfg)['train']
        cfg_train.update(epochs=iterations, learning_rate=self.lr, verbose=self.verbose)
        if self._track_window_on_device(optimize_kfs, iterations, cfg_train):
            return
        Trainer(cfg_train, self.grid, self.loss_fn, self.train_loader, None, self.cfg['device'],
                torch.float32).train()

    def _track_window_on_device(self, optimize_kfs, iterations, cfg_train) -> bool:
        """The window above for ONE keyframe as `iterations` library calls (ops.TrackAdamWindow) and one host
        synchronisation at the end: what Trainer.train() does here is, per epoch, one batch, MisoLossTracking through
        autograd (a unique() with a host read-back, so3_exp_map chains, the rigid map and its backward), a NaN check that
        reads the loss back, and Adam over the pose tensors of which only this keyframe's six numbers have a gradient --
        2.3 ms per iteration at 16 384 samples against ~0.1 ms of GPU work.  False: not this configuration, the
        Trainer runs."""
        from miso_amd import ops
        from miso_amd.grid_opt.loss import MisoLossTracking
        lf, grid = self.loss_fn, self.grid
        if (not self.fused or len(optimize_kfs) != 1 or type(lf) is not MisoLossTracking or lf.loss_type not in ('L1', 'L2', 'GM')
                or cfg_train.get('optimizer') != 'adam' or cfg_train.get('eval_every', -1) > 0
                or cfg_train.get('ckpt_every', -1) > 0 or cfg_train.get('pretrained_model') is not None
                or not hasattr(grid, '_fused_decoder') or iterations < 1):
            return False
        if str(self.cfg['device']).startswith('cpu') or not grid.Rwk.is_cuda:
            return False
        pack = grid._fused_decoder()
        if pack is None:
            return False
        kf_key = optimize_kfs[0]
        kf = grid.pose_key_to_id(f'KF{kf_key}')
        loader = self.train_loader
        batches = []
        win = None
        feats = [g.feature for g in grid.features]
        meta = grid.features[0].grid_meta(grid.ignore_level_)
        dr, dt = grid.rotation_corrections.data[kf], grid.translation_corrections.data[kf]
        dirs = self.__dict__.setdefault('_log_dirs_made', set())
        if cfg_train['log_dir'] not in dirs:                    # Trainer.set_logging creates them
            import os
            for d in (cfg_train['log_dir'], os.path.join(cfg_train['log_dir'], 'ckpt'),
                      os.path.join(cfg_train['log_dir'], 'tensorboard')):
                utils.cond_mkdir(d)
            dirs.add(cfg_train['log_dir'])
        grid.train()
        # one Adam step per BATCH per epoch: the step-scalar table and the loss ring of the window are sized for all of
        # them (a loader with several batches per epoch would otherwise run past the table -- its last row would be
        # reused, i.e. wrong bias corrections -- and finish() would cut the losses short)
        try:
            n_steps = int(iterations) * max(len(loader), 1)
        except TypeError:
            return False                                        # a loader without a length: let the Trainer run
        for epoch in range(iterations):
            for step_i, (model_input, gt) in enumerate(utils.iter_batches(loader)):
                model_input, gt = utils.prepare_batch(model_input, gt, self.cfg['device'], sanitize=False)
                coords_frame = model_input['coords_frame'][0]
                n = coords_frame.shape[0]
                if win is None:
                    win = self.__dict__.get('_adam_dev')
                    if (win is None or win.n != n or win.hyper != (float(self.lr), (0.9, 0.999), 1e-8, n_steps)
                            or win.pose.device != coords_frame.device):
                        win = self.__dict__['_adam_dev'] = ops.TrackAdamWindow(n, coords_frame.device, pack, self.lr,
                                                                               n_steps)
                    win.reset()
                elif n != win.n:
                    raise RuntimeError("tracking batches of one window differ in size")
                win.step(feats, meta, pack, coords_frame.contiguous(), gt['sdf'][0], gt['sdf_valid'][0],
                         model_input['sample_frame_ids'][0], kf_key, lf.trunc_dist, grid.Rwk[kf], grid.twk[kf], dr, dt,
                         lf.loss_type, lf.weight_sdf, lf.gm_scale_sdf, sanitize=True)
                batches.append((epoch, step_i))
        torch.autograd.graph.increment_version(grid.rotation_corrections)      # written through raw pointers:
        torch.autograd.graph.increment_version(grid.translation_corrections)   # pose caches key on the versions
        if win is None:
            return True
        losses, steps, skipped = win.finish()
        for _ in range(skipped):
            logger.warning("Loss is nan! Skip backward step.")
        if self.verbose:
            for (epoch, step_i), loss in zip(batches, losses):
                if step_i % 10 == 0:
                    logger.info(f"Train epoch {epoch} step {step_i} | train_loss={loss:.2e}.")
        return True

    def track(self, optimize_kf: int):
        if self.disable:
            return
        if self.solver == 'adam':
            self.track_window([optimize_kf], iterations=15)
        elif self.solver == 'lm':
            self.track_lm(optimize_kf)
        else:
            raise ValueError(f"Unknown solver: {self.solver}.")

    def track_lm(self, optimize_kf: int):
        for _ in range(self.lm_max_iter):
            info = self.lm_step(optimize_kf)
            if info['delta_R_deg'] < self.lm_tol_deg and info['delta_t_norm'] < self.lm_tol_m:
                break
        self.latest_fov_overlap = info['fov_overlap']

    def _lm_step_on_device(self, optimize_kf, coords_frame, frame_ids, gt_sdf, gt_valid):
        """The step below as one library call and one host synchronisation (ops.LmTrackStep) when the model takes the
        fused decoder path; None otherwise.  The reference's op-by-op version waits for the device eight times per
        step (nonzero, two asserts, the overlap count, the solve, three norms): 1.08 ms per step at 16 384 samples of
        which the GPU works for under 0.1 ms."""
        if (not self.fused or not coords_frame.is_cuda or self.loss_type not in ('L2', 'GM')
                or not hasattr(self.grid, '_fused_decoder')):
            return None
        pack = self.grid._fused_decoder()
        if pack is None or coords_frame.shape[0] == 0:
            return None
        from miso_amd import ops
        n = coords_frame.shape[0]
        step = self.__dict__.get('_lm_dev')
        if step is None or step.n != n or step.pose.device != coords_frame.device:
            step = self.__dict__['_lm_dev'] = ops.LmTrackStep(n, coords_frame.device, pack)
        grid = self.grid
        kf = grid.pose_key_to_id(f'KF{optimize_kf}')
        feats = [g.feature for g in grid.features]
        meta = grid.features[0].grid_meta(grid.ignore_level_)
        dr, dt = grid.rotation_corrections.data[kf], grid.translation_corrections.data[kf]
        try:
            out = step(feats, meta, pack, coords_frame.contiguous(), gt_sdf, gt_valid, frame_ids, optimize_kf,
                       self.trunc_dist, grid.Rwk[kf], grid.twk[kf], dr, dt, self.loss_type, self.gm_scale_sdf,
                       self.lm_lambda, sanitize=True)
        except ValueError:
            return None
        torch.autograd.graph.increment_version(grid.rotation_corrections)      # written through raw pointers:
        torch.autograd.graph.increment_version(grid.translation_corrections)   # pose caches key on the versions
        d_r, d_t, g_norm, n_in, n_keep, bad_frame, invalid = out[:7]
        assert bad_frame == 0
        assert invalid == 0, "Only valid SDFs should be used for tracking."
        return {'delta_R_deg': math.degrees(d_r), 'delta_t_norm': d_t, 'grad_norm': g_norm,
                'fov_overlap': float(n_in) / n_keep}

    def residual_weights(self, r: torch.Tensor):
        if self.loss_type == 'L2':
            return torch.ones_like(r)
        if self.loss_type == 'GM':
            return self.gm_scale_sdf / (self.gm_scale_sdf + r ** 2) ** 2
        raise ValueError(f"Unknown loss type: {self.loss_type}.")

    def lm_step(self, optimize_kf: int):
        """One damped Gauss-Newton step on the keyframe pose (reference :148-212):
        J_i = [ (hat(R x_i) grad_i)^T R , grad_i^T ], H = J^T W J + lambda I, g = J^T W r."""
        self.dataset.select_keyframes([optimize_kf])
        # (the device-side step takes the batch before nan_to_num and does that itself; any other path sanitises here)
        model_input, gt = utils.get_batch(self.train_loader, self.cfg['device'], sanitize=not self.fused)
        info = None
        if self.fused:
            info = self._lm_step_on_device(optimize_kf, model_input['coords_frame'][0], model_input['sample_frame_ids'][0],
                                           gt['sdf'][0], gt['sdf_valid'][0])
            if info is not None:
                return info
            model_input, gt = utils.sanitize_tensor_dict(model_input), utils.sanitize_tensor_dict(gt)
        coords_frame = model_input['coords_frame'][0]
        frame_ids = model_input['sample_frame_ids'][0]
        gt_sdf, gt_valid = gt['sdf'][0], gt['sdf_valid'][0]
        if self.trunc_dist is not None:
            keep = torch.nonzero(torch.abs(gt_sdf[:, 0]) < self.trunc_dist, as_tuple=False).squeeze(1)
            coords_frame, frame_ids = coords_frame[keep], frame_ids[keep]
            gt_sdf, gt_valid = gt_sdf[keep], gt_valid[keep]
        assert torch.all(frame_ids == optimize_kf)
        assert torch.all(gt_valid == 1), "Only valid SDFs should be used for tracking."
        R, t = self.grid.updated_kf_pose_from_key(f'KF{optimize_kf}')
        R, t = R.detach(), t.detach()
        coords_world = utils_geometry.transform_points_to(coords_frame, R, t)
        in_bound = utils_geometry.coords_in_bound(coords_world, self.grid.bound)
        fov_overlap = float(torch.count_nonzero(in_bound)) / in_bound.numel()
        # SDF value and spatial gradient from ONE forward + coordinate backward (no autograd graph)
        sdf_pred, grad_world = self.grid.sdf_and_gradient(coords_world)
        if coords_frame.is_cuda:
            # J, H = J^T W J and g = J^T W r in one launch
            from miso_amd import ops
            H, g, _ = ops.lm_normal_eq(coords_frame, R, grad_world, sdf_pred, gt_sdf, self.loss_type,
                                       self.gm_scale_sdf)
            H = H + self.lm_lambda * torch.eye(6, device=H.device)
        else:
            Rx = utils_geometry.transform_points_to(coords_frame, R, torch.zeros_like(t))
            cT = torch.bmm(hat(Rx), grad_world.unsqueeze(-1)).squeeze(-1)
            J = torch.cat((cT @ R, grad_world), dim=1)                     # (N,6) = [J_R, J_t]
            r = (sdf_pred - gt_sdf).detach()
            w = self.residual_weights(r)
            H = J.T @ (w * J) + self.lm_lambda * torch.eye(6, device=J.device)
            g = J.T @ (w * r)
        delta = torch.linalg.solve(H, -g)
        delta_R, delta_t = delta[:3], delta[3:]
        with torch.no_grad():
            kf = self.grid.pose_key_to_id(f'KF{optimize_kf}')
            self.grid.rotation_corrections[kf] += delta_R.squeeze()
            self.grid.translation_corrections[kf] += delta_t
        return {'delta_R_deg': math.degrees(torch.linalg.norm(delta_R).item()),
                'delta_t_norm': torch.linalg.norm(delta_t).item(),
                'grad_norm': torch.linalg.norm(g).item(), 'fov_overlap': fov_overlap}
