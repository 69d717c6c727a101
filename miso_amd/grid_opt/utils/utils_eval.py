"""Trajectory error metrics the alignment demo prints (reference: grid_opt/utils/utils_eval.py:110-147, which wraps the
``evo`` package).  ``evo`` is not a dependency here: the absolute pose error after a rigid (Umeyama, no scale)
alignment is ~40 lines of numpy.  Same call signature and the same ``get_all_statistics()`` keys as evo's APE.
The chamfer / F-score mesh metrics of the reference (Open3D nearest-neighbour queries on sampled meshes) are
evaluation tooling outside the hot path."""
import enum

import numpy as np

from . import utils_geometry


class PoseRelation(enum.Enum):
    """The members of evo.core.metrics.PoseRelation the reference uses (demo/align_submaps.py:134-137)."""
    full_transformation = "full transformation"
    translation_part = "translation part"
    rotation_part = "rotation part"
    rotation_angle_rad = "rotation angle in radians"
    rotation_angle_deg = "rotation angle in degrees"


def _relation_name(pose_relation) -> str:
    return getattr(pose_relation, "name", str(pose_relation))      # ours or evo's enum: same member names


def umeyama_rigid(src: np.ndarray, dst: np.ndarray):
    """R (3,3), t (3,) minimising sum |R src_i + t - dst_i|^2 (Umeyama 1991 without scale)."""
    mu_s, mu_d = src.mean(axis=0), dst.mean(axis=0)
    cov = (dst - mu_d).T @ (src - mu_s) / src.shape[0]
    U, _, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    return R, mu_d - R @ mu_s


class APE:
    """Absolute pose error of a trajectory against a reference one, per pose; statistics like evo's APE."""

    def __init__(self, pose_relation=PoseRelation.translation_part):
        self.pose_relation = pose_relation
        self.error = np.zeros(0)

    def process_data(self, data):
        ref, est = data                                           # lists of 4x4
        rel = _relation_name(self.pose_relation)
        errs = []
        for P, Q in zip(ref, est):
            E = np.linalg.inv(P) @ Q
            if rel == "translation_part":
                errs.append(np.linalg.norm(E[:3, 3]))
            elif rel == "rotation_part":
                errs.append(np.linalg.norm(E[:3, :3] - np.eye(3)))
            elif rel == "full_transformation":
                errs.append(np.linalg.norm(E - np.eye(4)))
            elif rel in ("rotation_angle_rad", "rotation_angle_deg"):
                ang = np.arccos(np.clip((np.trace(E[:3, :3]) - 1.0) / 2.0, -1.0, 1.0))
                errs.append(np.degrees(ang) if rel.endswith("deg") else ang)
            else:
                raise ValueError(f"unsupported pose relation {self.pose_relation}")
        self.error = np.asarray(errs, dtype=np.float64)

    def get_all_statistics(self):
        e = self.error
        return {"rmse": float(np.sqrt(np.mean(e ** 2))), "mean": float(np.mean(e)), "median": float(np.median(e)),
                "std": float(np.std(e)), "min": float(np.min(e)), "max": float(np.max(e)), "sse": float(np.sum(e ** 2))}


def get_evo_trajectory(R, t):
    """(n,3,3), (n,3[,1]) tensors -> list of 4x4 numpy poses (the reference returns an evo PosePath3D)."""
    return [utils_geometry.pose_matrix(R[i], t[i].reshape(3, 1)).detach().cpu().numpy().astype(np.float64)
            for i in range(R.shape[0])]


def evo_trajectory_error(R1, t1, R2, t2, pose_relation=PoseRelation.translation_part, align: bool = True) -> APE:
    """Reference :125-147: optionally align trajectory 2 to trajectory 1 rigidly, then the absolute pose error."""
    path1, path2 = get_evo_trajectory(R1, t1), get_evo_trajectory(R2, t2)
    if align:
        Ra, ta = umeyama_rigid(np.stack([P[:3, 3] for P in path2]), np.stack([P[:3, 3] for P in path1]))
        A = np.eye(4)
        A[:3, :3], A[:3, 3] = Ra, ta
        path2 = [A @ P for P in path2]
    ape = APE(pose_relation)
    ape.process_data((path1, path2))
    return ape
