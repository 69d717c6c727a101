"""Pure-PyTorch CPU restatement of MISO's encode/decode hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The product package
``miso_amd`` never imports this module; ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg use it as the checker.

Parity status
-------------
* ``encode_stock`` / ``sdf_stock`` are the reference's own op sequence
  (``normalize_coordinates`` -> per-level ``F.grid_sample`` -> ``cat`` ->
  ``nn.Sequential``) restated with stock torch ops; they are pinned against the
  imported reference through the golden vectors in ``tests/golden`` (generated
  by ``tools/make_goldens.py`` in the build container).
* ``trilinear_gather`` is an explicit 8-corner restatement that is differentiable
  to any order (ATen has no double backward for ``grid_sampler_3d``); it is
  pinned against ``F.grid_sample`` (value + first derivatives), against
  ``torch.autograd.gradgradcheck`` in fp64, and against the known-answer inputs
  of the reference's ``third_party/cuda_gridsample_grad2/test3d.py:17-35``.
* ``so3_exp_map`` / ``hat`` restate pytorch3d (un-vendored, unpinned
  ``git+https://github.com/facebookresearch/pytorch3d.git`` in the reference's
  ``environment.yaml:114``).  pytorch3d is absent from the image, so that
  boundary is **parity unpinned**: it is pinned only by our own goldens.

Every function cites the reference file:line (relative to the MISO repo) it
follows.  All functions are dtype-generic (fp32 for parity, fp64 for
gradcheck / error measurement).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# Coordinates
# --------------------------------------------------------------------------- #
def normalize_coordinates(x: torch.Tensor, bound: torch.Tensor) -> torch.Tensor:
    """grid_opt/utils/utils.py:22-51 -- map metres to [-1, 1] per axis."""
    bmin = bound[:, 0].view(1, -1)
    bmax = bound[:, 1].view(1, -1)
    return 2 * (x - bmin) / (bmax - bmin) - 1


def denormalize_coordinates(xn: torch.Tensor, bound: torch.Tensor) -> torch.Tensor:
    """grid_opt/utils/utils.py:53-82."""
    bmin = bound[:, 0].view(1, -1)
    bmax = bound[:, 1].view(1, -1)
    return (xn + 1) / 2 * (bmax - bmin) + bmin


def _unnormalize(c: torch.Tensor, size: int, align_corners: bool) -> torch.Tensor:
    """ATen ``grid_sampler_unnormalize`` (the op behind grid_modules.py:86-94)."""
    if align_corners:
        return ((c + 1) / 2) * (size - 1)
    return ((c + 1) * size - 1) / 2


# --------------------------------------------------------------------------- #
# Trilinear sampling
# --------------------------------------------------------------------------- #
def grid_sample_stock(feature: torch.Tensor, xn: torch.Tensor,
                      align_corners: bool = False,
                      padding_mode: str = "zeros") -> torch.Tensor:
    """The reference's exact call shape, grid_opt/models/grid_modules.py:86-94.

    feature (1,C,Z,Y,X), xn (N,3) normalised -> (N,C).
    """
    n = xn.shape[0]
    out = F.grid_sample(feature, xn.reshape(1, n, 1, 1, 3), mode="bilinear",
                        align_corners=align_corners, padding_mode=padding_mode)
    return out[0, :, :, 0, 0].transpose(0, 1)


def trilinear_gather(feature: torch.Tensor, xn: torch.Tensor,
                     align_corners: bool = False,
                     padding_mode: str = "zeros") -> torch.Tensor:
    """Explicit 8-corner trilinear sample, differentiable to any order.

    Same semantics as ATen ``grid_sampler_3d`` (bilinear): weights are
    ``(i0+1-ix)`` / ``(ix-i0)`` products, out-of-range corners contribute zero
    (``zeros``) or coordinates are clipped first (``border``); corner naming and
    weights as in third_party/cuda_gridsample_grad2/gridsample_cuda.cu:302-342.
    """
    assert feature.ndim == 5 and feature.shape[0] == 1
    assert padding_mode in ("zeros", "border")
    _, c, d, h, w = feature.shape
    flat = feature.reshape(c, d * h * w)
    coords = []
    for axis, size in ((0, w), (1, h), (2, d)):
        i = _unnormalize(xn[:, axis], size, align_corners)
        if padding_mode == "border":
            i = torch.clamp(i, 0, size - 1)
        coords.append(i)
    ix, iy, iz = coords
    x0 = torch.floor(ix).detach()
    y0 = torch.floor(iy).detach()
    z0 = torch.floor(iz).detach()
    out = 0
    for dz in (0, 1):
        wz = (iz - z0) if dz else (z0 + 1 - iz)
        zi = z0 + dz
        for dy in (0, 1):
            wy = (iy - y0) if dy else (y0 + 1 - iy)
            yi = y0 + dy
            for dx in (0, 1):
                wx = (ix - x0) if dx else (x0 + 1 - ix)
                xi = x0 + dx
                inb = ((xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
                       & (zi >= 0) & (zi < d))
                lin = (zi.clamp(0, d - 1) * h + yi.clamp(0, h - 1)) * w + xi.clamp(0, w - 1)
                vals = flat[:, lin.long()].transpose(0, 1)           # (N,C)
                wgt = (wx * wy * wz) * inb.to(feature.dtype)
                out = out + vals * wgt.unsqueeze(1)
    return out


def encode_stock(features: Sequence[torch.Tensor], bound: torch.Tensor,
                 x: torch.Tensor,
                 ignore_level: Optional[Sequence[bool]] = None) -> torch.Tensor:
    """grid_opt/utils/utils.py:143-164 (level loop + cat) with
    FeatureGrid.interpolate (grid_modules.py:72-95) inlined, stock ATen ops."""
    feats = []
    for lvl, f in enumerate(features):
        xn = normalize_coordinates(x, bound.to(x))
        v = grid_sample_stock(f, xn)
        if ignore_level is not None and ignore_level[lvl]:
            v = torch.zeros_like(v)
        feats.append(v)
    return torch.cat(feats, dim=1)


def encode_gather(features: Sequence[torch.Tensor], bound: torch.Tensor,
                  x: torch.Tensor,
                  ignore_level: Optional[Sequence[bool]] = None) -> torch.Tensor:
    """Same as ``encode_stock`` on the any-order-differentiable restatement."""
    feats = []
    for lvl, f in enumerate(features):
        xn = normalize_coordinates(x, bound.to(x))
        v = trilinear_gather(f, xn)
        if ignore_level is not None and ignore_level[lvl]:
            v = torch.zeros_like(v)
        feats.append(v)
    return torch.cat(feats, dim=1)


# --------------------------------------------------------------------------- #
# Decoder
# --------------------------------------------------------------------------- #
def mlp_forward(feats: torch.Tensor, weights: Sequence[torch.Tensor],
                biases: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
    """grid_opt/models/modules.py:16-21,31-32: Linear/ReLU chain, no activation
    after the last Linear.  ``weights[i]`` is (out,in) like ``nn.Linear.weight``."""
    h = feats
    last = len(weights) - 1
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = F.linear(h, w, b)
        if i != last:
            h = torch.relu(h)
    return h


def decoder_params(state_dict) -> Tuple[List[torch.Tensor], List[Optional[torch.Tensor]]]:
    """Split an ``MLPNet`` state-dict (keys ``network.{0,2,4,..}.{weight,bias}``,
    modules.py:16-21) into ordered weight / bias lists."""
    idx = sorted({int(k.split(".")[1]) for k in state_dict if k.startswith("network.")})
    ws = [state_dict[f"network.{i}.weight"] for i in idx]
    bs = [state_dict.get(f"network.{i}.bias") for i in idx]
    return ws, bs


def sdf_stock(features, bound, x, weights, biases, ignore_level=None):
    """GridNet.forward, grid_opt/models/grid_net.py:306-325 (pos_invariant)."""
    return mlp_forward(encode_stock(features, bound, x, ignore_level), weights, biases)


def sdf_gather(features, bound, x, weights, biases, ignore_level=None):
    return mlp_forward(encode_gather(features, bound, x, ignore_level), weights, biases)


# --------------------------------------------------------------------------- #
# Losses on the hot path
# --------------------------------------------------------------------------- #
def miso_loss_regression(pred, targ, valid_mask=None, sample_weights=None, loss_type="L1"):
    """grid_opt/loss.py:594-635 -- mean over ALL rows including masked ones."""
    n = pred.shape[0]
    if valid_mask is None:
        valid_mask = torch.ones((n, 1)).to(pred)
    if sample_weights is None:
        sample_weights = torch.ones((n, 1)).to(pred)
    if loss_type == "L2":
        v = torch.sum((pred - targ) ** 2, dim=1, keepdim=True)
    elif loss_type == "L1":
        v = torch.sum(torch.abs(pred - targ), dim=1, keepdim=True)
    elif loss_type == "Cosine":
        v = 1.0 - F.cosine_similarity(pred, targ, dim=1, eps=1e-8).unsqueeze(1)
    else:
        raise ValueError(loss_type)
    v = torch.where(valid_mask == 1, v, torch.zeros_like(v))
    return torch.mean(sample_weights * v)


def miso_loss_free_space(pred_sdf, gt_sdf, gt_sdf_sign, trunc_dist):
    """grid_opt/loss.py:668-700."""
    up = torch.where(gt_sdf_sign == 1, F.relu(pred_sdf - gt_sdf), torch.zeros_like(pred_sdf))
    lo = torch.where(gt_sdf_sign == 1, F.relu(trunc_dist - pred_sdf), torch.zeros_like(pred_sdf))
    return torch.mean(torch.maximum(up, lo))


# --------------------------------------------------------------------------- #
# Rigid-body maps (pose-Jacobian path)
# --------------------------------------------------------------------------- #
def hat(v: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.so3.hat restated (parity unpinned, see header):
    (B,3) -> (B,3,3) skew matrices."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    o = torch.zeros_like(x)
    return torch.stack([torch.stack([o, -z, y], -1),
                        torch.stack([z, o, -x], -1),
                        torch.stack([-y, x, o], -1)], -2)


def so3_exp_map(log_rot: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """pytorch3d.transforms.so3_exp_map restated (parity unpinned): Rodrigues
    with theta = sqrt(clamp(|w|^2, eps)).  Call sites in the reference:
    grid_opt/utils/utils_geometry.py:99, grid_opt/models/grid_net.py:7."""
    nrms = (log_rot * log_rot).sum(1)
    theta = torch.clamp(nrms, eps).sqrt()
    inv = 1.0 / theta
    fac1 = inv * theta.sin()
    fac2 = inv * inv * (1.0 - theta.cos())
    k = hat(log_rot)
    k2 = torch.bmm(k, k)
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return fac1[:, None, None] * k + fac2[:, None, None] * k2 + eye


def apply_pose_correction(R, t, R_delta, t_delta):
    """grid_opt/utils/utils_geometry.py:78-99: (R Exp(dr), t + dt)."""
    return torch.matmul(R, so3_exp_map(R_delta)[0]), t + t_delta


def transform_points_to(points_src, R_dst_src, t_dst_src):
    """grid_opt/utils/utils_geometry.py:214-225."""
    return points_src @ R_dst_src.T + t_dst_src.T


def transfrom_points_from(points_dst, R_dst_src, t_dst_src):
    """grid_opt/utils/utils_geometry.py:227-240 (sic: upstream spelling)."""
    return transform_points_to(points_dst, R_dst_src.T, -R_dst_src.T @ t_dst_src)


def transform_by_keyframe_loop(coords_frame, frame_ids, R_all, t_all):
    """The reference's per-keyframe loop (grid_opt/loss.py:763-774; same shape in loss_isdf.py:52-61 and
    align/miso.py:44-53): for every keyframe id present, the rows carrying it are mapped by
    transform_points_to with that keyframe's pose.  R_all (K,3,3), t_all (K,3,1)."""
    out = coords_frame.clone()
    for k in torch.unique(frame_ids).tolist():
        rows = torch.nonzero(frame_ids == k, as_tuple=False).squeeze(1)
        out[rows] = transform_points_to(coords_frame[rows], R_all[k], t_all[k])
    return out


def coords_in_bound(coords, bound):
    """grid_opt/utils/utils_geometry.py:11-27 -- inclusive box test, (N,1) bool."""
    return ((coords >= bound[:, 0]) & (coords <= bound[:, 1])).all(dim=1).unsqueeze(1)


def pairwise_latent_loss(feats_src, bound_src, feats_dst, bound_dst, coords_from,
                         R_src, t_src, R_dst, t_dst, level, fdim,
                         align_weight=3000.0, align_loss="L2", encode=encode_stock):
    """grid_opt/align/miso.py:116-211 on plain tensors (use_bound=True, no
    stability / truncation pruning, no subsampling)."""
    end_ch = fdim * (level + 1)
    world = transform_points_to(coords_from, R_src, t_src)
    coords_to = transfrom_points_from(world, R_dst, t_dst)
    mask = coords_in_bound(coords_to, bound_dst.to(coords_to))
    if torch.count_nonzero(mask) == 0:
        return torch.tensor(0)
    idx = torch.nonzero(mask, as_tuple=False)[:, 0]
    p_from = coords_from[idx]
    p_to = coords_to[idx]
    f_from = encode(feats_src, bound_src, p_from)[:, :end_ch]
    f_to = encode(feats_dst, bound_dst, p_to)[:, :end_ch]
    diff = f_from - f_to
    if align_loss == "L2":
        return torch.mean(diff ** 2) * align_weight
    if align_loss == "L1":
        return torch.mean(torch.linalg.vector_norm(diff, dim=1)) * align_weight
    raise ValueError(align_loss)


def lm_normal_equations(coords_frame, R, grad_world, sdf_pred, sdf_gt, loss_type="L2", gm_scale=0.1):
    """J (N,6), H = J^T W J, g = J^T W r of Tracker.lm_step (grid_opt/slam/tracker.py:176-196;
    weights :139-146; hat = pytorch3d.transforms.so3.hat, restated in `hat` above)."""
    Rx = coords_frame @ R.T
    cT = torch.bmm(hat(Rx), grad_world.unsqueeze(-1)).squeeze(-1)
    J = torch.cat((cT @ R, grad_world), dim=1)
    r = (sdf_pred - sdf_gt).reshape(-1, 1)
    if loss_type == "L2":
        w = torch.ones_like(r)
    elif loss_type == "GM":
        w = gm_scale / (gm_scale + r ** 2) ** 2
    else:
        raise ValueError(loss_type)
    return J, J.T @ (w * J), J.T @ (w * r)


# --------------------------------------------------------------------------- #
# Sample generation (the step that feeds the path).  Random draws are inputs.
# --------------------------------------------------------------------------- #
def ray_dirs_camera(H: int, W: int, fx, fy, cx, cy) -> torch.Tensor:
    """grid_opt/utils/utils_sample.py:10-30, depth_type 'z': (H,W,3) directions ((c-cx)/fx, (r-cy)/fy, 1)."""
    cols = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    rows = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    return torch.stack(((cols - cx) / fx, (rows - cy) / fy, torch.ones(H, W)), dim=-1)


def rgbd_sdf_samples(depth, T_WC, R_wk, t_wk, intrinsics, pix_b, pix_h, pix_w, u, g, *, min_depth,
                     dist_behind_surf, trunc_dist, n_strat, n_surf, normals=None, frame_ids=None):
    """PosedSdfRgbd.getitem_sdf, grid_opt/datasets/sdf_rgbd.py:381-483, with the random draws passed in.

    ``u`` (>= n1, n_strat) and ``g`` (>= n1, n_surf-1) are consumed by the n1 rays that survive the first filter
    (the reference draws them after that filter, utils_sample.py:241,284).  Returns the two dictionaries of
    :472-481 plus ``pc_world``, ``z_vals`` and the two ray counts."""
    B, H, W = depth.shape
    dirs = ray_dirs_camera(H, W, *intrinsics)
    d = depth[pix_b, pix_h, pix_w].reshape(-1)                      # utils_sample.py:156-157
    keep = d != 0
    if normals is not None:                                         # :160-166
        keep = keep & ~torch.isnan(normals[pix_b, pix_h, pix_w, 0])
    d, b, h, w = d[keep], pix_b[keep], pix_h[keep], pix_w[keep]
    n1 = d.numel()
    T = T_WC[b]
    dc = dirs[h, w]
    dw = (T[:, :3, :3] * dc[:, None, :]).sum(dim=-1)                # origin_dirs_W, :33-38
    org = T[:, :3, 3]
    far = d + dist_behind_surf                                      # sdf_rgbd.py:268
    span = (far - min_depth)[:, None]                               # stratified_sample, utils_sample.py:212-222
    edges = torch.linspace(0, 1, n_strat + 1)[None, :].repeat(n1, 1) * span + min_depth
    z = edges[:, :-1] + u[:n1] * (span / n_strat)                   # :241-244
    if n_surf == 1:                                                 # :278-281
        z = torch.cat((d[:, None], z), dim=1)
    elif n_surf > 1:                                                # :283-297
        near = torch.clamp(d[:, None] + g[:n1], torch.full((n1, 1), float(min_depth)), far[:, None])
        z = torch.cat((d[:, None], near, z), dim=1)
    pc = org[:, None, :] + dw[:, None, :] * z[:, :, None]           # :300
    sdf = dc.norm(dim=-1)[:, None] * (d[:, None] - z)               # bounds_ray, sdf_rgbd.py:525-528
    ok = ~torch.isnan(pc).reshape(n1, -1).any(dim=1)                # :416-424
    pc, sdf, b, z = pc[ok], sdf[ok], b[ok], z[ok]
    S = z.shape[1]
    ids = (b if frame_ids is None else frame_ids[b])[:, None].expand(-1, S).reshape(-1)   # :427-432
    world = pc.reshape(-1, 3)
    rb = b[:, None].expand(-1, S).reshape(-1)
    Rk, tk = R_wk[rb], t_wk.reshape(-1, 3)[rb]
    t_inv = -(Rk.transpose(1, 2) @ tk[:, :, None])[:, :, 0]         # transfrom_points_from, utils_geometry.py:227-240
    coords = (world[:, None, :] @ Rk)[:, 0, :] + t_inv              # x R + t_inv^T   (:436-445)
    sdf = sdf.reshape(-1, 1)
    sign = torch.zeros_like(sdf)                                    # :452-455
    sign[sdf < -trunc_dist] = -1
    sign[sdf > trunc_dist] = 1
    inputs = {"coords_frame": coords, "sample_frame_ids": ids[:, None].long(), "weights": torch.ones_like(sdf)}
    gt = {"sdf": sdf, "sdf_valid": sdf.abs() < trunc_dist, "sdf_signs": sign}
    return inputs, gt, {"pc_world": world, "z_vals": z, "n_first": n1, "n_kept": int(ok.sum())}


def lidar_distance_weight(dists, max_range, scale=0.8):
    """PosedSdf3DLidar.distance_weight_func, grid_opt/datasets/sdf_3d_lidar.py:205-211."""
    return 1 + scale * 0.5 - (dists / max_range) * scale


def lidar_frame_samples(pts_world, R_wf, t_wf, g_near, u_free, u_behind, *, near_surface_n, near_surface_std,
                        free_space_n, behind_surface_n, trunc_dist, min_dist_ratio, max_range):
    """One frame of PosedSdf3DLidar.sample_frames, grid_opt/datasets/sdf_3d_lidar.py:214-347 (after the
    sub-sampling permutation :233-237), draws passed in: g_near (n*near_n,1) standard normals, u_free
    (n*free_n,1) and u_behind (n*behind_n,1) uniforms.  fp64 like the reference's numpy, cast at :340-345."""
    p = pts_world.double()
    eye = t_wf.reshape(1, 3).double()
    dist = (p - eye).norm(dim=1, keepdim=True)                       # :240
    pts, sdfs, wts, sgn = [p], [torch.zeros_like(dist)], [lidar_distance_weight(dist, max_range)], [torch.zeros_like(dist)]

    def along(rep, d_new):
        rp = p.repeat_interleave(rep, dim=0)
        direc = rp - eye
        direc = direc / (direc.norm(dim=1, keepdim=True) + 1e-8)
        return eye + direc * d_new

    if near_surface_n > 0:                                           # :252-265
        rd = dist.repeat_interleave(near_surface_n, dim=0)
        dn = rd + g_near.double() * near_surface_std
        pts.append(along(near_surface_n, dn))
        sdfs.append((rd - dn).float().double())
        wts.append(lidar_distance_weight(rd, max_range))
        sgn.append(torch.zeros_like(rd))
    if free_space_n > 0:                                             # :270-285
        rd = dist.repeat_interleave(free_space_n, dim=0)
        span = torch.clamp((1.0 - trunc_dist / rd) - min_dist_ratio, min=1e-2)
        disp = ((min_dist_ratio + u_free.double() * span) - 1.0) * rd
        pts.append(along(free_space_n, rd + disp))
        sdfs.append((-disp.float()).double())
        wts.append(torch.ones_like(rd))
        sgn.append(torch.ones_like(rd))
    if behind_surface_n > 0:                                         # :290-302
        rd = dist.repeat_interleave(behind_surface_n, dim=0)
        disp = near_surface_std + u_behind.double() * (4 * near_surface_std - 2 * near_surface_std)
        pts.append(along(behind_surface_n, rd + disp))
        sdfs.append((-disp.float()).double())
        wts.append(torch.ones_like(rd))
        sgn.append(-torch.ones_like(rd))
    world = torch.cat(pts).float()                                   # :307-323
    sdf = torch.cat(sdfs).float()
    frame = transfrom_points_from(world, R_wf.float(), t_wf.reshape(3, 1).float())
    return {"points_frame": frame, "points_world_gt": world, "sdfs": sdf,
            "sdfs_valid": (sdf.abs() < trunc_dist).float(), "signs": torch.cat(sgn).float(),
            "weights": torch.cat(wts).float()}
