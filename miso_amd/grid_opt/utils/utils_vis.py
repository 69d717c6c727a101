"""Colour helpers of the reference's visualisation utilities (grid_opt/utils/utils_vis.py:14-40) that the demos use
without a window.  Importable without Open3D; the geometry builders of the reference (coordinate frames, line
meshes, trajectory plots) are Open3D GUI code outside the hot path and raise here with a clear message."""
import numpy as np


def beautiful_rgb():
    return [[1, 0.706, 0], [0, 0.651, 0.929], [0.17, 0.63, 0.17], [0.58, 0.40, 0.74], [0.12, 0.65, 0.65],
            [0.84, 0.15, 0.16]]


def convert_to_colormap(v: np.ndarray, cmap_name='seismic', thresh=0.10) -> np.ndarray:
    """Values (N,) -> RGB (N,3) on a symmetric [-thresh, thresh] colour scale."""
    import matplotlib.pyplot as plt
    from matplotlib import colors
    norm = colors.Normalize(vmin=-abs(thresh), vmax=abs(thresh), clip=True)
    return plt.get_cmap(cmap_name)(norm(np.asarray(v)))[:, :3]


def _needs_open3d(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"utils_vis.{name} builds Open3D geometry for the interactive viewer; it is not part "
                                  "of the MI355X hot path (install open3d and use the reference's utils_vis for plots)")
    fn.__name__ = name
    return fn


for _n in ("create_coordinate_frame", "create_lineset_from_numpy_traj", "create_spheres_from_numpy_traj",
           "visualize_submaps", "visualize_trajectories"):
    globals()[_n] = _needs_open3d(_n)
