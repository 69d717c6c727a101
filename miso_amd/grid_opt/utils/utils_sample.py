"""Ray / pixel sampling helpers (reference: grid_opt/utils/utils_sample.py).

The per-iteration work of the reference's ``get_batch_data`` / ``stratified_sample`` / ``sample_along_rays``
(:142-302) lives in the HIP kernel behind ``miso_amd.ops.sample_rays``; what stays here is set-up that runs once
per dataset (ray directions, back-projection, normals) and the pixel draw."""
import torch


def ray_dirs_C(B, H, W, fx, fy, cx, cy, device, depth_type='z'):
    """(B,H,W,3) camera-frame directions ((c-cx)/fx, (r-cy)/fy, 1); unit length for 'euclidean' (reference :10-30)."""
    cols = torch.arange(W, device=device, dtype=torch.float32)[None, :].expand(H, W)
    rows = torch.arange(H, device=device, dtype=torch.float32)[:, None].expand(H, W)
    dirs = torch.stack(((cols - cx) / fx, (rows - cy) / fy, torch.ones(H, W, device=device)), dim=-1)
    if depth_type == 'euclidean':
        dirs = dirs * (1. / dirs.norm(dim=-1))[..., None]
    return dirs[None].expand(B, H, W, 3).contiguous()


def origin_dirs_W(T_WC, dirs_C):
    """World-frame ray origins and directions (reference :33-38)."""
    return T_WC[:, :3, -1], (T_WC[:, :3, :3] * dirs_C[..., None, :]).sum(dim=-1)


def pointcloud_from_depth_torch(depth, fx, fy, cx, cy, depth_type="z", skip=1):
    """Back-project a (H,W) depth image to (H,W,3) camera-frame points; NaN depth stays NaN (reference :41-68)."""
    assert depth_type in ("z", "euclidean"), "Unexpected depth_type"
    H, W = depth.shape
    cols = torch.arange(0, W, skip, device=depth.device)[None, :]
    rows = torch.arange(0, H, skip, device=depth.device)[:, None]
    z = depth[::skip, ::skip]
    pc = torch.stack((z * (cols - cx) / fx, z * (rows - cy) / fy, z), dim=-1)
    if depth_type == "euclidean":
        pc = pc * (z / pc.norm(dim=-1))[..., None]
    return pc


def estimate_pointcloud_normals(points):
    """Per-pixel normals of an organised (H,W,3) cloud (reference :71-126): of the 8 neighbour pairs two pixels
    away and a quarter turn apart, take the pair closest to the anchor and cross its two edge vectors.  Pixels
    whose chosen edges are NaN / degenerate give NaN normals (that NaN is what the ray filter reads)."""
    assert points.shape[2] == 3
    d = 2
    H, W = points.shape[:2]
    nan = float('nan')
    padded = torch.full((H + 2 * d, W + 2 * d, 3), nan, device=points.device, dtype=points.dtype)
    padded[d:d + H, d:d + W] = points
    # neighbour k (row, col) offsets, counter-clockwise from "left"
    ring = [(-d, 0), (-d, d), (0, d), (d, d), (d, 0), (d, -d), (0, -d), (-d, -d)]
    # the reference indexes lookups[k] as (i offset, j offset) with i = row
    nb = torch.stack([padded[d + a:d + a + H, d + b:d + b + W] for a, b in ring])        # (8,H,W,3)
    e2 = nb - points[None]
    e3 = nb.roll(-2, dims=0) - points[None]
    cost = e2.norm(dim=-1) + e3.norm(dim=-1)
    cost = torch.where(torch.isnan(cost), torch.full_like(cost, float('inf')), cost)
    best = cost.argmin(dim=0)[None, :, :, None].expand(1, H, W, 3)
    n = torch.cross(e2.gather(0, best)[0], e3.gather(0, best)[0], dim=-1)
    return n / n.norm(dim=-1, keepdim=True)


def sample_pixels(n_rays, n_frames, h, w, device):
    """n_rays uniform pixels in each of n_frames frames (reference :129-139)."""
    total = n_rays * n_frames
    indices_h = torch.randint(0, h, (total,), device=device)
    indices_w = torch.randint(0, w, (total,), device=device)
    indices_b = torch.arange(n_frames, device=device).repeat_interleave(n_rays)
    return indices_b, indices_h, indices_w
