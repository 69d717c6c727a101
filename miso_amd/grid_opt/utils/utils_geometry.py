"""Rigid-body maps of the pose path (reference: grid_opt/utils/utils_geometry.py)."""
import math

import numpy as np
import torch

from miso_amd.so3 import so3_exp_map


def coords_in_bound(coords: torch.Tensor, bound: torch.Tensor) -> torch.Tensor:
    """Inclusive box test, (N,3) x (3,2) -> bool (N,1) (reference :11-27; the debug
    count_nonzero of the reference, a host sync, is dropped)."""
    inside = (coords >= bound[:, 0]) & (coords <= bound[:, 1])
    return inside.all(dim=1, keepdim=True)


def identity_rotations(n: int) -> torch.Tensor:
    return torch.eye(3).unsqueeze(0).repeat(n, 1, 1)


def pose_matrix(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    assert R.shape == (3, 3) and t.shape == (3, 1)
    T = torch.eye(4).to(R)
    T[:3, :3] = R
    T[:3, 3:] = t
    return T


def apply_pose_correction(R, t, R_delta, t_delta):
    """(R Exp(dr), t + dt) with dr (1,3), dt (3,1) (reference :78-99)."""
    assert R.shape == (3, 3) and t.shape == (3, 1)
    assert R_delta.shape == (1, 3) and t_delta.shape == (3, 1)
    return R @ so3_exp_map(R_delta)[0], t + t_delta


def transform_points_to(points_src, R_dst_src, t_dst_src):
    """Rows of points_src mapped by x -> R x + t (reference :214-225)."""
    assert R_dst_src.shape == (3, 3) and t_dst_src.shape == (3, 1)
    return points_src @ R_dst_src.T + t_dst_src.T


def transfrom_points_from(points_dst, R_dst_src, t_dst_src):
    """Inverse map x -> R^T (x - t) (reference :227-240; upstream spelling kept)."""
    assert R_dst_src.shape == (3, 3) and t_dst_src.shape == (3, 1)
    return transform_points_to(points_dst, R_dst_src.T, -(R_dst_src.T @ t_dst_src))


def transform_poses_to(R_src_frames, t_src_frames, R_dst_src, t_dst_src):
    """Compose poses with a frame change (reference :242-261)."""
    assert R_dst_src.shape == (3, 3) and t_dst_src.shape == (3, 1)
    return R_dst_src @ R_src_frames, R_dst_src @ t_src_frames + t_dst_src


def transform_poses_from(R_dst_frames, t_dst_frames, R_dst_src, t_dst_src):
    """Poses given in dst expressed in src: compose with the inverse frame change (reference :263-278)."""
    assert R_dst_src.shape == (3, 3) and t_dst_src.shape == (3, 1)
    return transform_poses_to(R_dst_frames, t_dst_frames, R_dst_src.T, -(R_dst_src.T @ t_dst_src))


def batch_transform_to_world_frame(coords_frame, frame_indices, R_world_frame_input, t_world_frame_input,
                                   rotation_corrections, translation_corrections):
    """Rows [begin, end) of every frame mapped to the world by that frame's corrected pose; frame 0 is the anchor
    and keeps its input pose (reference :29-59).  One batched product instead of a loop with a cat."""
    F = R_world_frame_input.shape[0]
    assert t_world_frame_input.shape[0] == F and rotation_corrections.shape[0] == F \
        and translation_corrections.shape[0] == F
    keep = torch.ones(F, 1, 1, dtype=coords_frame.dtype, device=coords_frame.device)
    keep[0] = 0.0
    eye = torch.eye(3, dtype=coords_frame.dtype, device=coords_frame.device)
    R_corr = so3_exp_map(rotation_corrections) * keep + eye * (1.0 - keep)
    R = R_world_frame_input @ R_corr
    t = t_world_frame_input + translation_corrections * keep
    spans = [(int(frame_indices[f, 0]), int(frame_indices[f, 1])) for f in range(F)]
    return torch.cat([coords_frame[b:e] @ R[f].T + t[f].T for f, (b, e) in enumerate(spans)], dim=0)


def matrix_to_axis_angle(R: torch.Tensor) -> torch.Tensor:
    """(B,3,3) -> (B,3) rotation vectors (what pytorch3d.transforms.matrix_to_axis_angle returns: axis * angle with
    the angle in [0, pi]).  Restated from the definition: angle from the trace, axis from the skew part, with the
    small-angle limit axis*angle -> vee(R - R^T)/2."""
    skew = torch.stack((R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]), dim=1) * 0.5
    cos = ((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] - 1.0) * 0.5).clamp(-1.0, 1.0)
    sin = skew.norm(dim=1)
    angle = torch.atan2(sin, cos)
    scale = torch.where(sin > 1e-6, angle / sin.clamp(min=1e-12), torch.ones_like(sin))
    return skew * scale.unsqueeze(1)


def get_pose_correction(R, t, Rnew, tnew):
    """(dr (1,3), dt (3,1)) with Rnew = R Exp(dr), tnew = t + dt (reference :101-119)."""
    return matrix_to_axis_angle((R.T @ Rnew).unsqueeze(0)), tnew - t


def _unit_rows(v):
    return v / (v.norm(dim=-1, keepdim=True) + 1e-8)          # utils.normalize_last_dim


def uniform_translations(k, bound):
    """(k,3) translations uniform in the box; numpy's global RNG, x then y then z columns (reference :121-125)."""
    cols = [np.random.uniform(bound[a, 0], bound[a, 1], k).reshape(k, 1) for a in range(3)]
    return torch.from_numpy(np.concatenate(cols, axis=1)).float()


def gaussian_translations(k, stddev):
    return torch.from_numpy(np.random.normal(size=(k, 3)) * stddev).float()


def fixed_length_translations(k, length):
    return length * _unit_rows(torch.from_numpy(np.random.normal(size=(k, 3))).float())


def wrapped_gaussian_rotations(n, std_rad=0.1):
    """(n,3,3) rotations Exp(w), w ~ N(0, std_rad^2 I) from numpy's global RNG (reference :140-146)."""
    return so3_exp_map(torch.from_numpy(np.random.normal(size=(n, 3)) * std_rad).float())


def fixed_angle_rotations(n, rad):
    return so3_exp_map(rad * _unit_rows(torch.from_numpy(np.random.normal(size=(n, 3))).float()))


def chordal_to_radian(d):
    return 2 * np.arcsin(d / (2 * np.sqrt(2)))


def chordal_to_degree(d):
    return math.degrees(chordal_to_radian(d))


def _relative_angles(R1, R2):
    """Angles (rad, numpy) of R1 R2^T from the trace (pytorch3d.so3_relative_angle with cos_angle=True, then the
    reference's clip + arccos, :169-170)."""
    cos = ((R1 * R2).sum(dim=(1, 2)) - 1.0) * 0.5
    return np.arccos(np.clip(cos.detach().cpu().numpy(), -1, 1))


def rotation_rmse(R1, R2):
    """Root mean squared angle between two (N,3,3) sets, degrees (reference :160-173)."""
    return math.degrees(np.sqrt(np.mean(_relative_angles(R1, R2) ** 2)))


def rotation_mean_error(R1, R2):
    return math.degrees(np.mean(np.abs(_relative_angles(R1, R2))))


def translation_rmse(t1, t2):
    """(N,3,1) x (N,3,1) -> root mean squared distance (reference :190-200)."""
    return torch.sqrt(torch.mean(torch.linalg.vector_norm(t1.squeeze() - t2.squeeze(), dim=1) ** 2)).item()


def translation_mean_error(t1, t2):
    return torch.mean(torch.linalg.vector_norm(t1.squeeze() - t2.squeeze(), dim=1)).item()


def aabb_torch(points: torch.Tensor, buffer: float = 0.0):
    """(3,2) [min - buffer, max + buffer] rows of an (N,3) cloud (reference :280-290)."""
    return torch.stack((points.min(dim=0)[0] - buffer, points.max(dim=0)[0] + buffer), dim=0).T


def voxel_down_sample_torch(points: torch.Tensor, voxel_size: float):
    """Indices of one point per occupied voxel: the one closest to the voxel centre after quantising the distance
    into 1000 steps, ties to the lowest index (reference :292-335, incl. its float voxel key).  Returned in the
    order of the sorted voxel keys."""
    steps = 1000
    cell = torch.floor(points / voxel_size)
    dist = ((points - (cell + 0.5) * voxel_size) ** 2).sum(dim=1) ** 0.5
    rank = (dist / dist.max() * (steps - 1)).long()
    ijk = cell.long() - torch.floor(points.min(dim=0)[0] / voxel_size).long()
    side = ijk.max().float().ceil()
    key = ijk[:, 0] + ijk[:, 1] * side + ijk[:, 2] * side * side
    _, inverse = torch.unique(key, return_inverse=True)
    n = points.shape[0]
    base = 10 ** len(str(n - 1))
    code = torch.arange(n, dtype=inverse.dtype, device=points.device) + rank * base
    best = torch.full((int(inverse.max()) + 1,), torch.iinfo(inverse.dtype).max, dtype=inverse.dtype,
                      device=points.device).scatter_reduce_(0, inverse, code, reduce="amin", include_self=True)
    return best % base


def crop_points(points: torch.Tensor, ts: torch.Tensor, min_z_th=-3.0, max_z_th=100.0, min_range=2.75,
                max_range=100.0):
    """Range and height gate of a lidar scan (strict inequalities); ts rows follow (reference :337-358)."""
    dist = torch.norm(points, dim=1)
    keep = (dist > min_range) & (dist < max_range) & (points[:, 2] > min_z_th) & (points[:, 2] < max_z_th)
    return points[keep], (ts[keep] if ts is not None else None)


def check_numpy_pose_matrix(T: np.ndarray):
    """Finite, 4x4, last row (0,0,0,1), det R = 1, R^T R = I to 1e-5 (reference :360-389)."""
    if np.isinf(T).any() or np.isnan(T).any() or T.shape != (4, 4) or not np.allclose(T[3, :], [0, 0, 0, 1]):
        return False
    R = T[:3, :3]
    return bool(np.allclose(np.linalg.det(R), 1) and np.allclose(R.T @ R, np.eye(3), atol=1e-5))


def read_kitti_format_poses(filename: str):
    """List of 4x4 float64 poses from a KITTI pose file (12 numbers per line, row-major 3x4); None if a line is
    too short (reference :391-413)."""
    poses = []
    with open(filename, 'r') as fh:
        for line in fh:
            vals = line.strip().split()
            if len(vals) < 12:
                return None
            T = np.eye(4)
            T[:3, :4] = np.array([float(v) for v in vals[:12]]).reshape(3, 4)
            poses.append(T)
    return poses


def write_kitti_format_poses(filename: str, poses_np: np.ndarray, direct_use_filename=False):
    """(N,4,4) -> one row-major 3x4 line per pose, np.savetxt's default format (reference :415-424)."""
    np.savetxt(fname=filename if direct_use_filename else f"{filename}_kitti.txt",
               X=poses_np[:, :3, :].reshape(poses_np.shape[0], -1))
