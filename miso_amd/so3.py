"""SO(3) helpers on the pose path.

pytorch3d (``so3_exp_map``, ``hat``) is an un-vendored, unpinned dependency of the
reference (environment.yaml:114; call sites grid_opt/utils/utils_geometry.py:99,
grid_opt/models/grid_net.py:7, grid_opt/slam/tracker.py:5) and is not in this
image, so its published formula is restated here: Rodrigues with the angle
clamped at sqrt(eps) (so d/dw at w = 0 is sin(0.01)/0.01, not 1).
"""
import torch


def hat(v: torch.Tensor) -> torch.Tensor:
    """(B,3) -> (B,3,3) skew-symmetric matrices, [v]_x u = v x u."""
    assert v.ndim == 2 and v.shape[1] == 3
    x, y, z = v.unbind(dim=1)
    zero = torch.zeros_like(x)
    rows = (torch.stack((zero, -z, y), dim=1),
            torch.stack((z, zero, -x), dim=1),
            torch.stack((-y, x, zero), dim=1))
    return torch.stack(rows, dim=1)


def so3_exp_map(log_rot: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """(B,3) axis-angle -> (B,3,3) rotation matrices."""
    assert log_rot.ndim == 2 and log_rot.shape[1] == 3
    sq = log_rot.pow(2).sum(dim=1)
    angle = sq.clamp(min=eps).sqrt()
    a = angle.sin() / angle
    b = (1.0 - angle.cos()) / (angle * angle)
    K = hat(log_rot)
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device).expand_as(K)
    return eye + a.view(-1, 1, 1) * K + b.view(-1, 1, 1) * (K @ K)
