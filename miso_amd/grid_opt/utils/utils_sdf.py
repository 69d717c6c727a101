"""Dense field extraction (reference: grid_opt/utils/utils_sdf.py:69-86).  The reference
evaluates res^3 queries in 16^3-point chunks -- 4 096 model calls and as many ``.cpu()`` syncs
at resolution 256.  Here the lattice is generated on the device slab by slab (a few million
points per call, forward-only fused encode+decode) and copied to the host once.

Mesh extraction (reference :89-140) runs on the volume where it lies: ``extract_geometry`` hands the
device-resident volume to the HIP marching cubes (``ops.marching_cubes``; the reference copies it to the
host for PyMCubes) and ``save_mesh`` writes the PLY itself -- mcubes, trimesh and open3d are not needed;
the returned ``TriangleMesh`` carries the arrays an open3d mesh would (vertices, triangles, vertex_normals)."""
import logging
import os

import numpy as np
import torch

import miso_amd.grid_opt.utils.utils as utils   # noqa: F401  (the demos reach `utils` through `from ...utils_sdf import *`)

logger = logging.getLogger(__name__)


def sign_mask_from_gt_sdf(gt_sdf: torch.Tensor, trunc_dist=0.15) -> torch.Tensor:
    """(N,1) labels: 1 where sdf > trunc_dist (free space), else 0 (reference :19-37), without the host syncs."""
    return (gt_sdf > trunc_dist).to(gt_sdf.dtype)


def valid_mask_from_gt_sdf(gt_sdf: torch.Tensor, trunc_dist=0.15) -> torch.Tensor:
    """(N,1) labels: 1 where |sdf| < trunc_dist (reference :40-58)."""
    return (gt_sdf.abs() < trunc_dist).to(gt_sdf.dtype)


def extract_fields(bound_min: torch.Tensor, bound_max: torch.Tensor, resolution, query_func,
                   device=None, max_points=1 << 22):
    """u[i,j,k] = query_func((x_i, y_j, z_k)) on the res^3 lattice spanned by linspace per axis,
    returned as a float32 numpy array of shape (res, res, res) like the reference."""
    return extract_fields_device(bound_min, bound_max, resolution, query_func, device, max_points).cpu().numpy()


def extract_fields_device(bound_min: torch.Tensor, bound_max: torch.Tensor, resolution, query_func,
                          device=None, max_points=1 << 22, lattice_func=None) -> torch.Tensor:
    """extract_fields with the (res, res, res) volume left on the device (what marching cubes reads).
    lattice_func(xs, ys, zs) -> (nx, ny, nz) volume or None: a model that evaluates a meshgrid lattice itself (GridAtlas.
    sdf_on_lattice: one launch per slab, the points generated inside the kernel from the three axis vectors); tried
    first, slabs of at most 2^30 points."""
    lo = bound_min.detach().cpu()
    hi = bound_max.detach().cpu()
    if device is None:
        device = "cuda:0" if torch.cuda.is_available() else "cpu"
    axes = [torch.linspace(float(lo[a]), float(hi[a]), resolution) for a in range(3)]   # host linspace, as upstream
    if lattice_func is not None:
        out = torch.empty((resolution, resolution, resolution), dtype=torch.float32, device=device)
        ys_d, zs_d = axes[1].to(device), axes[2].to(device)
        slab = max(1, (1 << 30) // (resolution * resolution))
        ok = True
        for x0 in range(0, resolution, slab):
            vol = lattice_func(axes[0][x0:x0 + slab].to(device), ys_d, zs_d)
            if vol is None:
                ok = False
                break
            out[x0:x0 + vol.shape[0]] = vol
        if ok:
            return out
    ys, zs = axes[1].to(device), axes[2].to(device)
    out = torch.empty((resolution, resolution, resolution), dtype=torch.float32, device=device)
    slab = max(1, max_points // (resolution * resolution))
    with torch.no_grad():
        for x0 in range(0, resolution, slab):
            xs = axes[0][x0:x0 + slab].to(device)
            xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing='ij')
            pts = torch.stack((xx, yy, zz), dim=-1).reshape(-1, 3)
            out[x0:x0 + xs.shape[0]] = query_func(pts).reshape(xs.shape[0], resolution, resolution)
    return out


def extract_geometry(bound_min: torch.Tensor, bound_max: torch.Tensor, resolution, threshold, query_func,
                     device=None, lattice_func=None):
    """Level set ``threshold`` of the field as (vertices (V,3) float64 in metres, triangles (T,3) int64), numpy
    (reference :89-101).  Vertices in index coordinates are mapped to the bound exactly as upstream:
    ``v / (res - 1) * (max - min) + min``."""
    from miso_amd import ops
    u = extract_fields_device(bound_min, bound_max, resolution, query_func, device, lattice_func=lattice_func)
    verts, tris = ops.marching_cubes(u, float(threshold))
    lo = bound_min.detach().cpu().numpy()
    hi = bound_max.detach().cpu().numpy()
    vertices = verts.cpu().numpy().astype(np.float64) / (resolution - 1.0) * (hi - lo)[None, :] + lo[None, :]
    return vertices, tris.cpu().numpy()


class TriangleMesh:
    """The arrays of the open3d mesh the reference's save_mesh returns."""

    def __init__(self, vertices, triangles):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.triangles = np.ascontiguousarray(triangles, dtype=np.int64).reshape(-1, 3)
        self.vertex_normals = None

    def apply_transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        self.vertices = self.vertices @ T[:3, :3].T + T[:3, 3][None, :]
        return self

    def compute_vertex_normals(self):
        """Area-weighted average of the incident face normals, normalised."""
        v, f = self.vertices, self.triangles
        fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        vn = np.zeros_like(v)
        for k in range(3):
            np.add.at(vn, f[:, k], fn)
        norm = np.linalg.norm(vn, axis=1, keepdims=True)
        self.vertex_normals = vn / np.where(norm > 0, norm, 1.0)
        return self

    def export_ply(self, path):
        """Binary little-endian PLY: float32 x y z per vertex, uchar-counted int32 index lists per face (what
        trimesh's ``export(file_type='ply')`` writes for a bare mesh)."""
        v = self.vertices.astype('<f4')
        f = np.empty(len(self.triangles), dtype=[('n', 'u1'), ('idx', '<i4', (3,))])
        f['n'] = 3
        f['idx'] = self.triangles
        header = ("ply\nformat binary_little_endian 1.0\n"
                  f"element vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
                  f"element face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n")
        with open(path, 'wb') as fh:
            fh.write(header.encode('ascii'))
            fh.write(v.tobytes())
            fh.write(f.tobytes())


def read_ply(path) -> TriangleMesh:
    """Reads back what TriangleMesh.export_ply wrote."""
    with open(path, 'rb') as fh:
        nv = nf = None
        while True:
            line = fh.readline().decode('ascii').strip()
            if line.startswith('element vertex'):
                nv = int(line.split()[-1])
            elif line.startswith('element face'):
                nf = int(line.split()[-1])
            elif line == 'end_header':
                break
        v = np.frombuffer(fh.read(12 * nv), dtype='<f4').reshape(nv, 3)
        f = np.frombuffer(fh.read(13 * nf), dtype=[('n', 'u1'), ('idx', '<i4', (3,))])
    return TriangleMesh(v, f['idx'])


def save_mesh(model, bounds: torch.Tensor, save_path=None, resolution=256, device='cuda:0', flip_face=True,
              transform: torch.Tensor = None) -> TriangleMesh:
    """Zero level set of ``model`` inside ``bounds`` ((3,2) rows [min,max]) as a triangle mesh, optionally moved by
    the 4x4 ``transform`` and written as PLY (reference :104-140)."""
    if save_path is not None:
        logger.info(f"Saving mesh to {save_path}...")
        os.makedirs(os.path.dirname(save_path), exist_ok=True)

    def query_func(pts):
        with torch.no_grad():
            return model(pts.to(device))

    vertices, triangles = extract_geometry(bounds[:, 0], bounds[:, 1], resolution=resolution, threshold=0,
                                           query_func=query_func, device=device,
                                           lattice_func=getattr(model, 'sdf_on_lattice', None))
    if flip_face:
        triangles = triangles[:, [2, 1, 0]]
    mesh = TriangleMesh(vertices, triangles)
    if transform is not None:
        mesh.apply_transform(transform.detach().cpu().numpy())
    if save_path is not None:
        mesh.export_ply(save_path)
    return mesh.compute_vertex_normals()
