"""Interface of the SLAM datasets (reference: grid_opt/datasets/submap_dataset.py)."""
from torch import Tensor
from torch.utils.data import Dataset


class SubmapDataset(Dataset):
    """A dataset of posed keyframes.  ``__getitem__`` returns ``(input_dict, gt_dict)`` with

    input_dict: 'coords_frame' (N,3) samples in their keyframe's frame, 'sample_frame_ids' (N,1) int64,
                'weights' (N,1);
    gt_dict:    'sdf' (N,1), 'sdf_valid' (N,1), 'sdf_signs' (N,1) in {-1 occupied, 0 within the truncation
                band, +1 free}                                                  (reference :57-76)."""

    @property
    def num_kfs(self) -> int:
        raise NotImplementedError

    def get_odometry_at_pose(self, src_id) -> Tensor:
        """4x4 odometry estimate from keyframe src_id to src_id + 1."""
        raise NotImplementedError

    def sampled_points_at_kf(self, kf_id) -> Tensor:
        """(N,3) sampled points of one keyframe, in its own frame."""
        raise NotImplementedError

    def select_keyframes(self, kf_ids):
        raise NotImplementedError

    def unselect_keyframes(self):
        raise NotImplementedError

    def true_kf_pose_in_world(self, kf_id):
        raise NotImplementedError

    def noisy_kf_pose_in_world(self, kf_id):
        raise NotImplementedError

    def __getitem__(self, index):
        raise NotImplementedError
