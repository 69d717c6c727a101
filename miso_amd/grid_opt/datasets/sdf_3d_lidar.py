"""SDF samples from posed LiDAR frames (reference: grid_opt/datasets/sdf_3d_lidar.py).

The reference builds every frame's samples once on the host with numpy (``sample_frames``, :214-347) and then, per
iteration, draws a subset per frame with ``np.random.choice`` and concatenates five indexed tensors per frame
(``getitem_world``, :374-428).  Here the samples of all frames are one packed device table
``[x y z | sdf valid sign weight]`` plus frame offsets, built with batched device ops, and a batch is one random
key sort plus one row gather -- no per-frame loop, nothing on the host.

Point-cloud files (.pcd / .ply through open3d, :100-102) and their voxel down-sampling are storage concerns
outside this path: frames enter through ``from_frames`` as sensor-frame point arrays.
"""
import logging

import numpy as np
import torch

from miso_amd.grid_opt.datasets.submap_dataset import SubmapDataset
from miso_amd.grid_opt.utils.utils_geometry import pose_matrix

logger = logging.getLogger(__name__)


def crop_points(points, ts, min_z_th=-3.0, max_z_th=100.0, min_range=2.75, max_range=100.0):
    """Range / height crop in the sensor frame (reference utils_geometry.py:337-358)."""
    dist = torch.norm(points, dim=1)
    keep = (dist > min_range) & (dist < max_range) & (points[:, 2] > min_z_th) & (points[:, 2] < max_z_th)
    return points[keep], (ts[keep] if ts is not None else None)


class PosedSdf3DLidar(SubmapDataset):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("reading .pcd/.ply frames needs open3d, which this build does not depend on; "
                                  "use PosedSdf3DLidar.from_frames(points_local, poses_gt, poses_init, ...)")

    @classmethod
    def from_frames(cls, points_local, poses_gt, poses_init=None, frame_batchsize=2 ** 10, frame_samples=2 ** 10,
                    near_surface_n=4, near_surface_std=0.1, free_space_n=2, behind_surface_n=1, trunc_dist=0.50,
                    distance_std=0., min_dist_ratio=0.30, min_z=-3.0, max_z=60.0, min_range=1.5, max_range=60.0,
                    crop=True, device='cuda:0', generator=None, draws=None):
        """points_local: list of (n_f,3) sensor-frame clouds; poses_*: (F,4,4) world poses.  ``draws`` (tests):
        per frame a dict {perm, g_near, u_free, u_behind} replacing the random draws of sample_frames."""
        self = cls.__new__(cls)
        SubmapDataset.__init__(self)
        self.frame_batchsize, self.frame_samples = frame_batchsize, frame_samples
        self.near_surface_n, self.near_surface_std = near_surface_n, near_surface_std
        self.free_space_n, self.behind_surface_n = free_space_n, behind_surface_n
        self.trunc_dist, self.distance_std, self.min_dist_ratio = trunc_dist, distance_std, min_dist_ratio
        self.max_range_hehind_surface = 4 * near_surface_std
        self.min_z, self.max_z, self.min_range, self.max_range = min_z, max_z, min_range, max_range
        self.device = device
        if distance_std > 1e-12:
            raise ValueError("Noise on surface points not implemented yet.")
        poses_gt = torch.as_tensor(np.asarray(poses_gt), dtype=torch.float32)
        poses_init = poses_gt if poses_init is None else torch.as_tensor(np.asarray(poses_init), dtype=torch.float32)
        self._num_frames = len(points_local)
        self.R_world_frame_gt = poses_gt[:self._num_frames, :3, :3].contiguous()
        self.t_world_frame_gt = poses_gt[:self._num_frames, :3, 3:].contiguous()
        self.R_world_frame = poses_init[:self._num_frames, :3, :3].contiguous()
        self.t_world_frame = poses_init[:self._num_frames, :3, 3:].contiguous()
        self.frames_lidar = []
        for f, pts in enumerate(points_local):
            pts = torch.as_tensor(np.asarray(pts), dtype=torch.float32)
            if crop:
                pts, _ = crop_points(pts, None, min_z, max_z, min_range, max_range)
            glob = pts.double() @ self.R_world_frame_gt[f].double().T + self.t_world_frame_gt[f].double().T
            self.frames_lidar.append({"points_local": pts, "points_global": glob})
        self._selected_kfs = None
        self._plan = {}
        self.sample_frames(generator=generator, draws=draws)
        return self

    @property
    def num_kfs(self):
        return self._num_frames

    def sampled_points_at_kf(self, kf_id):
        return self.frames_data[kf_id]['points_frame']

    def get_odometry_at_pose(self, src_id):
        T_src = pose_matrix(*self.noisy_kf_pose_in_world(src_id))
        T_dst = pose_matrix(*self.noisy_kf_pose_in_world(src_id + 1))
        return torch.linalg.inv(T_src) @ T_dst

    def distance_weight_func(self, dists, dist_weight_scale=0.8):
        """PIN-SLAM style range weight (reference :205-211)."""
        return 1 + dist_weight_scale * 0.5 - (dists / self.max_range) * dist_weight_scale

    def true_kf_pose_in_world(self, kf_id):
        return self.R_world_frame_gt[kf_id], self.t_world_frame_gt[kf_id]

    def noisy_kf_pose_in_world(self, kf_id):
        return self.R_world_frame[kf_id], self.t_world_frame[kf_id]

    def __len__(self):
        return 1

    def select_keyframes(self, kf_ids):
        self._selected_kfs = list(kf_ids)

    def unselect_keyframes(self):
        self._selected_kfs = None

    # ------------------------------------------------------------------ one-time sample construction
    def sample_frames(self, generator=None, draws=None):
        """Surface, near-surface, free-space and behind-surface samples of every frame (reference :214-347).
        fp64 on the device where the reference uses numpy doubles; stored as fp32 like :340-345."""
        dev = self.device
        rows, sizes = [], []
        self.frames_data = []
        for f in range(self._num_frames):
            eye = self.t_world_frame_gt[f].reshape(1, 3).to(dev).double()
            surf = self.frames_lidar[f]["points_global"].to(dev)
            n = surf.shape[0]
            keep = min(self.frame_samples, n)
            d = None if draws is None else draws[f]
            perm = (torch.randperm(n, device=dev, generator=generator) if d is None
                    else torch.as_tensor(d["perm"], device=dev))[:keep]
            surf = surf[perm]
            dist = (surf - eye).norm(dim=1, keepdim=True)

            def draw(kind, count, normal=False):
                if d is not None:
                    return torch.as_tensor(d[kind], device=dev, dtype=torch.float64).reshape(count, 1)
                fn = torch.randn if normal else torch.rand
                return fn(count, 1, device=dev, dtype=torch.float64, generator=generator)

            def shoot(rep, new_dist):
                direction = surf.repeat_interleave(rep, dim=0) - eye
                direction = direction / (direction.norm(dim=1, keepdim=True) + 1e-8)
                return eye + direction * new_dist

            pts, sdf = [surf], [torch.zeros_like(dist)]
            wgt, sgn = [self.distance_weight_func(dist)], [torch.zeros_like(dist)]
            if self.near_surface_n > 0:
                rd = dist.repeat_interleave(self.near_surface_n, dim=0)
                dn = rd + draw("g_near", rd.shape[0], normal=True) * self.near_surface_std
                pts.append(shoot(self.near_surface_n, dn))
                sdf.append(rd - dn)
                wgt.append(self.distance_weight_func(rd))
                sgn.append(torch.zeros_like(rd))
            if self.free_space_n > 0:
                rd = dist.repeat_interleave(self.free_space_n, dim=0)
                span = torch.clamp((1.0 - self.trunc_dist / rd) - self.min_dist_ratio, min=1e-2)
                disp = ((self.min_dist_ratio + draw("u_free", rd.shape[0]) * span) - 1.0) * rd
                pts.append(shoot(self.free_space_n, rd + disp))
                sdf.append(-disp)
                wgt.append(torch.ones_like(rd))
                sgn.append(torch.ones_like(rd))
            if self.behind_surface_n > 0:
                rd = dist.repeat_interleave(self.behind_surface_n, dim=0)
                disp = self.near_surface_std + draw("u_behind", rd.shape[0]) * (
                    self.max_range_hehind_surface - 2 * self.near_surface_std)
                pts.append(shoot(self.behind_surface_n, rd + disp))
                sdf.append(-disp)
                wgt.append(torch.ones_like(rd))
                sgn.append(-torch.ones_like(rd))
            world = torch.cat(pts).float()
            sdf = torch.cat(sdf).float()
            Rf = self.R_world_frame_gt[f].to(dev)
            frame = world @ Rf + (-(Rf.T @ self.t_world_frame_gt[f].to(dev))).T      # transfrom_points_from
            valid = (sdf.abs() < self.trunc_dist).float()
            table = torch.cat([frame, sdf, valid, torch.cat(sgn).float(), torch.cat(wgt).float()], dim=1)
            rows.append(table)
            sizes.append(table.shape[0])
            self.frames_data.append({"points_frame": table[:, 0:3], "points_world_gt": world, "sdfs": table[:, 3:4],
                                     "sdfs_valid": table[:, 4:5], "signs": table[:, 5:6], "weights": table[:, 6:7]})
        self._table = torch.cat(rows) if rows else torch.zeros(0, 7, device=dev)
        self._sizes = sizes
        self._starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    # ------------------------------------------------------------------ per-iteration batch
    def _batch_plan(self):
        """Static index arithmetic of the current keyframe selection (cached)."""
        kfs = list(range(self._num_frames))
        if self._selected_kfs is not None:
            kfs = list(set(kfs).intersection(self._selected_kfs))            # reference :385-387 (set order)
        key = tuple(kfs)
        plan = self._plan.get(key)
        if plan is None:
            dev = self.device
            src = torch.cat([torch.arange(self._starts[f], self._starts[f + 1]) for f in kfs]) if kfs else \
                torch.zeros(0, dtype=torch.int64)
            seg = torch.cat([torch.full((self._sizes[f],), i, dtype=torch.float64) for i, f in enumerate(kfs)]) if kfs \
                else torch.zeros(0, dtype=torch.float64)
            take, ids, base = [], [], 0
            for f in kfs:
                k = min(self.frame_batchsize, self._sizes[f])
                take.append(torch.arange(base, base + k))
                ids.append(torch.full((k, 1), f, dtype=torch.int64))
                base += self._sizes[f]
            plan = (src.to(dev), seg.to(dev), torch.cat(take).to(dev) if take else src.to(dev),
                    torch.cat(ids).to(dev) if ids else torch.zeros(0, 1, dtype=torch.int64, device=dev))
            self._plan = {key: plan}
        return plan

    def getitem_world(self, index, generator=None, choice=None):
        """frame_batchsize rows of every selected frame, drawn without replacement (reference :374-428).
        ``choice`` (tests): per selected frame the row indices to take, as np.random.choice returned them."""
        src, seg, take, ids = self._batch_plan()
        if choice is None:
            # a uniformly random order inside every frame's segment: sort (segment + U[0,1)) keys
            keys = seg + torch.rand(seg.shape[0], device=self.device, dtype=torch.float64, generator=generator)
            rows = src[torch.argsort(keys)[take]]
        else:
            kfs = [int(f) for f in ids[:, 0].unique_consecutive().tolist()]
            rows = torch.cat([torch.as_tensor(c, device=self.device) + int(self._starts[f])
                              for f, c in zip(kfs, choice)])
        batch = self._table[rows]
        input_dict = {'coords_frame': batch[:, 0:3], 'sample_frame_ids': ids, 'weights': batch[:, 6:7]}
        gt_dict = {'sdf': batch[:, 3:4], 'sdf_valid': batch[:, 4:5], 'sdf_signs': batch[:, 5:6]}
        return input_dict, gt_dict

    def __getitem__(self, index):
        return self.getitem_world(index)
