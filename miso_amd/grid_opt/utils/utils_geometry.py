"""Rigid-body maps of the pose path (reference: grid_opt/utils/utils_geometry.py)."""
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
