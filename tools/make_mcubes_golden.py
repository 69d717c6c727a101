"""Golden vectors for marching cubes from an independent third-party implementation.

PyMCubes (what the reference calls, grid_opt/utils/utils_sdf.py:94) is not in this image; scikit-image 0.18.3 under
/opt/conda is, and its ``marching_cubes(..., method='lorensen')`` is the same published algorithm.  Run with

    /opt/conda/bin/python3.9 tools/make_mcubes_golden.py

It writes tests/golden/mcubes.npz: for each test volume the volume itself, the iso level, the unique vertices
(sorted lexicographically), the face count, the surface area and the absolute enclosed volume.  Vertex sets do not
depend on the case table; areas/volumes/face counts do only through ambiguous cells.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def volumes():
    out = {}
    n = 28
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n - 4), np.arange(n + 3), indexing="ij"), -1).astype(np.float64)
    out["sphere"] = ((np.linalg.norm(g - np.array([13.3, 11.6, 14.9]), axis=-1) - 8.7).astype(np.float32), 0.0)
    q = g - np.array([13.5, 11.5, 15.0])
    out["torus"] = ((np.sqrt((np.sqrt(q[..., 0] ** 2 + q[..., 2] ** 2) - 8.2) ** 2 + q[..., 1] ** 2) - 3.1)
                    .astype(np.float32), 0.05)
    s = g * 0.45
    out["gyroid"] = ((np.sin(s[..., 0]) * np.cos(s[..., 1]) + np.sin(s[..., 1]) * np.cos(s[..., 2])
                      + np.sin(s[..., 2]) * np.cos(s[..., 0])).astype(np.float32), 0.13)
    rs = np.random.RandomState(5)
    out["noise"] = (rs.standard_normal((14, 11, 13)).astype(np.float32), 0.21)       # ambiguous cells galore
    return out


def main():
    from skimage.measure import marching_cubes, mesh_surface_area
    out = {}
    for name, (u, iso) in volumes().items():
        v, f, _, _ = marching_cubes(u, level=iso, method="lorensen", allow_degenerate=True)
        v = np.unique(v.astype(np.float32), axis=0)
        vv, ff, _, _ = marching_cubes(u, level=iso, method="lorensen")
        a, b, c = (vv[ff[:, i]].astype(np.float64) for i in range(3))
        nrm = np.cross(b - a, c - a)
        out[f"{name}_u"] = u
        out[f"{name}_iso"] = np.float32(iso)
        out[f"{name}_verts"] = v
        out[f"{name}_faces"] = np.int64(len(ff))
        out[f"{name}_area"] = np.float64(0.5 * np.linalg.norm(nrm, axis=1).sum())
        out[f"{name}_volume"] = np.float64(abs((a * nrm).sum() / 6.0))
        print(name, u.shape, "verts", len(v), "faces", len(ff), "area", out[f"{name}_area"], "vol", out[f"{name}_volume"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mcubes.npz"), **out)


if __name__ == "__main__":
    main()
