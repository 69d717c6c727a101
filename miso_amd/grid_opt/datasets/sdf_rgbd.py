"""SDF samples from a posed RGB-D sequence (reference: grid_opt/datasets/sdf_rgbd.py).

Same constructor, attributes and ``__getitem__`` contract as the reference's ``PosedSdfRgbd``; the frames live on
the device and every ``__getitem__`` is one call of the HIP sampler (``miso_amd.ops.sample_rays``) instead of the
reference's ~40 tensor ops, two compactions and a per-keyframe Python loop (:381-483).  ``sample_batch`` is the
same thing without the read-back of the row count: a fixed-capacity batch whose tail is neutral padding.

File formats read by the constructor (reference :153-219): ``frames/pose/{i}.pose.txt`` (4x4 text),
``frames/depth/{i}.depth.pgm`` (16-bit PGM, parsed here without OpenCV), optional ``poses_color_icp.txt``
(KITTI rows).  Colour images are only used by the reference's CLIP branch, which is outside this path.
"""
import logging
import os
from os.path import join

import numpy as np
import torch

from miso_amd import ops
from miso_amd.grid_opt.datasets.submap_dataset import SubmapDataset
from miso_amd.grid_opt.utils import utils_geometry
from miso_amd.grid_opt.utils.utils_sample import (estimate_pointcloud_normals, pointcloud_from_depth_torch,
                                                  ray_dirs_C, sample_pixels)

logger = logging.getLogger(__name__)


def read_pgm16(path) -> np.ndarray:
    """Binary PGM (P5), 8 or 16 bit big-endian -> (H,W) integer array."""
    with open(path, 'rb') as f:
        data = f.read()
    fields, pos = [], 0
    while len(fields) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b'#':
            pos = data.index(b'\n', pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        fields.append(data[pos:end])
        pos = end
    if fields[0] != b'P5':
        raise ValueError(f"{path}: not a binary PGM")
    w, h, maxval = int(fields[1]), int(fields[2]), int(fields[3])
    dt = np.dtype('>u2') if maxval > 255 else np.dtype('u1')
    return np.frombuffer(data, dtype=dt, count=w * h, offset=pos + 1).reshape(h, w)


def read_kitti_format_poses(filename):
    """Rows of 12 numbers = the top 3x4 of each pose (reference utils_geometry.py:391-413)."""
    rows = np.loadtxt(filename, ndmin=2)
    if rows.shape[1] < 12:
        return None
    out = np.tile(np.eye(4), (rows.shape[0], 1, 1))
    out[:, :3, :] = rows[:, :12].reshape(-1, 3, 4)
    return list(out)


class PosedSdfRgbd(SubmapDataset):
    def __init__(self, dataset_root: str, num_input_frames: int, cam_params, min_depth=0.07, max_depth=12.0,
                 voxel_size=None, n_rays=2 ** 10, dist_behind_surf=0.1, n_strat_samples=3, n_surf_samples=4,
                 trunc_dist=0.30, frame_downsample=1, device='cuda:0', use_clip=False, padded=False):
        super().__init__()
        self.padded = padded
        if use_clip:
            raise NotImplementedError("the CLIP branch of the reference dataset is outside this path")
        self._configure(cam_params, min_depth, max_depth, voxel_size, n_rays, dist_behind_surf, n_strat_samples,
                        n_surf_samples, trunc_dist, device)
        self.dataset_root = dataset_root
        self._num_total = num_input_frames
        self.frame_downsample = frame_downsample
        idxs = range(0, num_input_frames, frame_downsample)
        poses = [np.loadtxt(join(dataset_root, 'frames', 'pose', f"{i}.pose.txt")).reshape(4, 4) for i in idxs]
        R_gt = torch.tensor(np.stack([p[:3, :3] for p in poses]), dtype=torch.float32)
        t_gt = torch.tensor(np.stack([p[:3, 3:] for p in poses]), dtype=torch.float32)
        icp = join(dataset_root, 'poses_color_icp.txt')
        R_est = t_est = None
        if os.path.exists(icp):
            T = np.stack(read_kitti_format_poses(icp))
            assert T.shape[0] == len(poses)
            R_est = torch.tensor(T[:, :3, :3], dtype=torch.float32)
            t_est = torch.tensor(T[:, :3, 3:], dtype=torch.float32)
        else:
            logger.warning(f"ICP pose file {icp} does not exist. Using GT instead.")
        depth = np.stack([read_pgm16(join(dataset_root, 'frames', 'depth', f"{i}.depth.pgm")).astype(np.float32)
                          * (1.0 / cam_params.depth_scale) for i in idxs])
        self._install_frames(torch.from_numpy(depth), R_gt, t_gt, R_est, t_est)

    @classmethod
    def from_frames(cls, depth_batch, R_world_frame_gt, t_world_frame_gt, cam_params, R_world_frame=None,
                    t_world_frame=None, min_depth=0.07, max_depth=12.0, voxel_size=None, n_rays=2 ** 10,
                    dist_behind_surf=0.1, n_strat_samples=3, n_surf_samples=4, trunc_dist=0.30, device='cuda:0',
                    normals=None, padded=False):
        """In-memory construction: (B,H,W) depth in metres and (B,3,3)/(B,3,1) keyframe poses."""
        self = cls.__new__(cls)
        SubmapDataset.__init__(self)
        self.padded = padded
        self._configure(cam_params, min_depth, max_depth, voxel_size, n_rays, dist_behind_surf, n_strat_samples,
                        n_surf_samples, trunc_dist, device)
        self._num_total = depth_batch.shape[0]
        self.frame_downsample = 1
        self._install_frames(depth_batch, R_world_frame_gt, t_world_frame_gt, R_world_frame, t_world_frame, normals)
        return self

    def _configure(self, cam_params, min_depth, max_depth, voxel_size, n_rays, dist_behind_surf, n_strat_samples,
                   n_surf_samples, trunc_dist, device):
        if voxel_size is not None:
            raise NotImplementedError("per-iteration voxel down-sampling (reference :462-470, a host round trip "
                                      "the reference itself marks FIXME) is not part of the device path")
        self._cam_params = cam_params
        self.min_depth, self.max_depth, self.voxel_size = min_depth, max_depth, voxel_size
        self.n_rays, self.dist_behind_surf = n_rays, dist_behind_surf
        self.n_strat_samples, self.n_surf_samples, self.trunc_dist = n_strat_samples, n_surf_samples, trunc_dist
        self.bounds_method, self.normal_trunc_dist = 'ray', 0.30
        self.device, self.use_clip = device, False
        self._selected_kfs = None
        self._frame_cache = {}
        self.dirs_C = None    # built on demand: the sampler derives directions from the intrinsics itself

    def _install_frames(self, depth, R_gt, t_gt, R_est=None, t_est=None, normals=None):
        c, dev = self._cam_params, self.device
        depth = depth.to(device=dev, dtype=torch.float32).clone()
        depth[depth > self.max_depth] = 0.                                # DepthFilter, utils_data.py:34-47
        self._num_frames = depth.shape[0]
        self.R_world_frame_gt = R_gt.to(torch.float32).cpu()
        self.t_world_frame_gt = t_gt.to(torch.float32).reshape(-1, 3, 1).cpu()
        self.R_world_frame = self.R_world_frame_gt.clone() if R_est is None else R_est.to(torch.float32).cpu()
        self.t_world_frame = self.t_world_frame_gt.clone() if t_est is None else t_est.reshape(-1, 3, 1).cpu()
        if normals is None:                                               # reference :205-207
            normals = torch.stack([estimate_pointcloud_normals(
                pointcloud_from_depth_torch(d, c.fx, c.fy, c.cx, c.cy)) for d in depth])
        T = torch.eye(4).repeat(self._num_frames, 1, 1)
        T[:, :3, :3] = self.R_world_frame_gt
        T[:, :3, 3:] = self.t_world_frame_gt
        self._depth_batch, self._norm_batch, self._T_WC_batch = depth, normals.to(dev), T.to(dev)
        self._R_gt_dev, self._t_gt_dev = self.R_world_frame_gt.to(dev), self.t_world_frame_gt.to(dev)

    def __len__(self):
        return 1

    @property
    def num_kfs(self) -> int:
        return self._num_frames

    def sampled_points_at_kf(self, kf_id):
        self.select_keyframes([kf_id])
        b = self.sample_batch()
        self.unselect_keyframes()
        return b.coords_frame[:b.rows()]

    def get_odometry_at_pose(self, src_id):
        T_src = utils_geometry.pose_matrix(*self.noisy_kf_pose_in_world(src_id))
        T_dst = utils_geometry.pose_matrix(*self.noisy_kf_pose_in_world(src_id + 1))
        return torch.linalg.inv(T_src) @ T_dst

    def select_keyframes(self, kf_ids):
        self._selected_kfs = list(kf_ids)

    def unselect_keyframes(self):
        self._selected_kfs = None

    def true_kf_pose_in_world(self, kf_id):
        return self.R_world_frame_gt[kf_id], self.t_world_frame_gt[kf_id]

    def noisy_kf_pose_in_world(self, kf_id):
        return self.R_world_frame[kf_id], self.t_world_frame[kf_id]

    # ------------------------------------------------------------------ sampling
    def _selected_frames(self):
        """Device views of the selected keyframes (gathered once per selection)."""
        key = None if self._selected_kfs is None else tuple(self._selected_kfs)
        hit = self._frame_cache.get(key)
        if hit is None:
            if key is None:
                ids = torch.arange(self._num_frames)
                hit = (self._depth_batch, self._T_WC_batch, self._norm_batch, self._R_gt_dev, self._t_gt_dev, ids)
            else:
                ids = torch.tensor(key, dtype=torch.int64)
                sel = ids.to(self.device)
                hit = (self._depth_batch[sel], self._T_WC_batch[sel], self._norm_batch[sel], self._R_gt_dev[sel],
                       self._t_gt_dev[sel], ids)
            c = self._cam_params
            sampler = ops.RaySampler(hit[0], hit[1], hit[3], hit[4], (c.fx, c.fy, c.cx, c.cy), normals=hit[2],
                                     frame_ids=hit[5], rays_per_frame=self.n_rays, min_depth=self.min_depth,
                                     dist_behind_surf=self.dist_behind_surf, trunc_dist=self.trunc_dist,
                                     n_strat=self.n_strat_samples, n_surf=self.n_surf_samples)
            hit = hit[:5] + (hit[5].to(self.device), sampler)
            self._frame_cache = {key: hit}
        return hit

    def sample_batch(self, out=None, keep_world=False, draws=None) -> ops.RayBatch:
        """One batch of n_rays pixels per selected keyframe -> RayBatch (capacity rays * S rows, live count on the
        device).  ``draws`` = (pix_h, pix_w, u, g) overrides the random draws (tests)."""
        depth, T_WC, norm, R, t, ids, sampler = self._selected_frames()
        c, nf = self._cam_params, depth.shape[0]
        total = self.n_rays * nf
        if draws is None:
            _, ph, pw = sample_pixels(self.n_rays, nf, c.H, c.W, self.device)
            u = torch.rand(total, self.n_strat_samples, device=self.device)
            g = torch.randn(total, max(self.n_surf_samples - 1, 0), device=self.device) * 0.1   # utils_sample.py:284
        else:
            ph, pw, u, g = draws
        return sampler(ph, pw, u, g, out=out, keep_world=keep_world)

    def sample_points(self, depth_batch=None, T_WC_batch=None, norm_batch=None, active_loss_approx=None):
        """World-frame samples of the current selection (reference :221-293): {'pc' (rays,S,3), 'z_vals'}."""
        if active_loss_approx is not None:
            raise Exception('Active sampling not currently supported.')
        b = self.sample_batch(keep_world=True)
        rays = b.rows() // b.S
        return {"pc": b.pc_world[:rays * b.S].reshape(rays, b.S, 3), "z_vals": b.z_vals[:rays * b.S].reshape(rays, b.S),
                "depth_batch": self._selected_frames()[0]}

    def getitem_sdf(self, index, draws=None):
        """Exact-size rows like the reference (one read-back of the row count), or with ``padded=True`` the whole
        fixed-capacity batch plus ``input_dict['live_rows']`` (int32 on the device): dropped rays leave neutral
        rows at the tail, nothing is read back and the batch shape never changes, so the trainer replays one
        captured step."""
        b = self.sample_batch(draws=draws)
        if self.padded:
            aux = b.aux
            input_dict = {'coords_frame': b.coords_frame, 'sample_frame_ids': b.sample_frame_ids[:, None],
                          'weights': aux[:, 3:4], 'live_rows': b.live_rows}
            return input_dict, {'sdf': aux[:, 0:1], 'sdf_valid': aux[:, 1:2] > 0, 'sdf_signs': aux[:, 2:3]}
        n = b.rows()
        aux = b.aux[:n]
        input_dict = {'coords_frame': b.coords_frame[:n], 'sample_frame_ids': b.sample_frame_ids[:n, None],
                      'weights': aux[:, 3:4]}
        gt_dict = {'sdf': aux[:, 0:1], 'sdf_valid': aux[:, 1:2] > 0, 'sdf_signs': aux[:, 2:3]}
        return input_dict, gt_dict

    def __getitem__(self, index):
        return self.getitem_sdf(index)

    def compute_scene_obb(self):
        """Bounding box of one batch of samples in the world frame (reference :495-515 returns Open3D's minimal
        oriented box of the voxel-down-sampled cloud; callers use its centre to aim a viewer).  Here: the
        axis-aligned box, with the accessors the demo calls."""
        model_input, _ = self.__getitem__(0)
        ids = model_input['sample_frame_ids'][:, 0].to(self.device)
        n = int(model_input.get('live_rows', torch.tensor(ids.shape[0])).item())
        x = model_input['coords_frame'][:n]
        world = torch.einsum('nij,nj->ni', self._R_gt_dev[ids[:n]], x) + self._t_gt_dev[ids[:n], :, 0]
        lo, hi = world.min(dim=0).values.cpu().numpy(), world.max(dim=0).values.cpu().numpy()
        return AxisAlignedBox(lo, hi)


class AxisAlignedBox:
    def __init__(self, lo, hi):
        self.min_bound, self.max_bound = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)

    def get_center(self):
        return 0.5 * (self.min_bound + self.max_bound)

    def get_extent(self):
        return self.max_bound - self.min_bound
