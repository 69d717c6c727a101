"""Dense field extraction (reference: grid_opt/utils/utils_sdf.py:69-86).  The reference
evaluates res^3 queries in 16^3-point chunks -- 4 096 model calls and as many ``.cpu()`` syncs
at resolution 256.  Here the lattice is generated on the device slab by slab (a few million
points per call, forward-only fused encode+decode) and copied to the host once.  Marching
cubes / mesh export stay out of scope (mcubes, trimesh, open3d are not on the hot path)."""
import numpy as np
import torch


def extract_fields(bound_min: torch.Tensor, bound_max: torch.Tensor, resolution, query_func,
                   device=None, max_points=1 << 22):
    """u[i,j,k] = query_func((x_i, y_j, z_k)) on the res^3 lattice spanned by linspace per axis,
    returned as a float32 numpy array of shape (res, res, res) like the reference."""
    lo = bound_min.detach().cpu()
    hi = bound_max.detach().cpu()
    if device is None:
        device = "cuda:0" if torch.cuda.is_available() else "cpu"
    axes = [torch.linspace(float(lo[a]), float(hi[a]), resolution) for a in range(3)]   # host linspace, as upstream
    ys, zs = axes[1].to(device), axes[2].to(device)
    out = torch.empty((resolution, resolution, resolution), dtype=torch.float32, device=device)
    slab = max(1, max_points // (resolution * resolution))
    with torch.no_grad():
        for x0 in range(0, resolution, slab):
            xs = axes[0][x0:x0 + slab].to(device)
            xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing='ij')
            pts = torch.stack((xx, yy, zz), dim=-1).reshape(-1, 3)
            out[x0:x0 + xs.shape[0]] = query_func(pts).reshape(xs.shape[0], resolution, resolution)
    return out.cpu().numpy()
