"""ScanNet scene metadata and the RGB-D dataset factory the demos call (reference:
grid_opt/utils/utils_scannet.py:10-113).  ``create_scannet_dataset`` builds the device-resident ``PosedSdfRgbd`` of
miso_amd.grid_opt.datasets.sdf_rgbd from the same files the reference reads (``<root>/scene<id>/scene<id>.txt``,
``frames/pose/*.pose.txt``, ``frames/depth/*.depth.pgm``).  The mesh-to-mesh ICP helper of the reference
(``align_mesh_to_ref``, an Open3D registration pipeline for evaluation plots) is outside the hot path."""
import logging
from dataclasses import dataclass
from os.path import join

from miso_amd.grid_opt.datasets.sdf_rgbd import PosedSdfRgbd
from miso_amd.grid_opt.utils.utils_data import CameraParameters

logger = logging.getLogger(__name__)


@dataclass
class SceneMetadata:
    bound: list
    name: str
    path: str
    intrinsics_file: str
    gt_mesh: str
    num_kfs: int
    anchor_kfs: list


# (bound, num_kfs, anchor_kfs) per scene, reference :21-64
_SCENES = {
    '0000_00': ([[-0.02, 10.38], [-0.01, 8.74], [-0.01, 3.03]], 372, [0, 124, 255]),
    '0011_00': ([[1.50, 7.50], [-0.05, 8.25], [-0.05, 2.70]], 159, [0, 73, 86, 121]),
    '0024_00': ([[0.00, 7.20], [-0.05, 8.05], [-0.05, 2.50]], 227, [0, 30, 84, 101, 131]),
    '0207_00': ([[1.00, 9.00], [0.00, 7.10], [-0.10, 2.90]], 133, [0, 35]),
}


def scannet_scenes():
    out = {}
    for sid, (bound, num_kfs, anchors) in _SCENES.items():
        root = f"./data/ScanNet/scene{sid}_mipsfusion"
        out[sid] = SceneMetadata(name=sid, path=root, intrinsics_file=f"{root}/scene{sid}.txt",
                                 gt_mesh=f"/home/hanwen/data/ScanNet/scans/scene{sid}/scene{sid}_vh_clean.ply",
                                 bound=[list(b) for b in bound], num_kfs=num_kfs, anchor_kfs=list(anchors))
    return out


def get_scannet_metadata(file):
    """``key = value`` lines of a ScanNet scene file -> dict of strings."""
    info = {}
    with open(file, 'r') as f:
        for line in f.read().splitlines():
            parts = line.split(' = ')
            if len(parts) == 2:
                info[parts[0]] = parts[1]
    return info


def get_scannet_cam_intrinsics(file) -> CameraParameters:
    info = get_scannet_metadata(file)
    return CameraParameters(depth_scale=1000.0, fx=float(info['fx_depth']), fy=float(info['fy_depth']),
                            cx=float(info['mx_depth']), cy=float(info['my_depth']), H=int(info['depthHeight']),
                            W=int(info['depthWidth']))


def create_scannet_dataset(scannet_root: str, scene_id: str, trunc_dist: float = 0.15, frame_downsample: int = 15,
                           n_rays: int = 200, n_surf_samples: int = 8, n_strat_samples: int = 19,
                           voxel_size: float = None, device='cuda:0', padded=False) -> PosedSdfRgbd:
    """Reference :85-113, same defaults.  ``device`` / ``padded`` are additions (the frames live on the device;
    padded batches let the trainer replay one captured step, see PosedSdfRgbd)."""
    scene_name = f"scene{scene_id}"
    scene_file = join(scannet_root, scene_name, f"{scene_name}.txt")
    info = get_scannet_metadata(scene_file)
    return PosedSdfRgbd(dataset_root=join(scannet_root, scene_name), num_input_frames=int(info['numColorFrames']),
                        cam_params=get_scannet_cam_intrinsics(scene_file), frame_downsample=frame_downsample,
                        n_rays=n_rays, min_depth=0.07, max_depth=12.0, n_surf_samples=n_surf_samples,
                        n_strat_samples=n_strat_samples, trunc_dist=trunc_dist, voxel_size=voxel_size, device=device,
                        padded=padded)
