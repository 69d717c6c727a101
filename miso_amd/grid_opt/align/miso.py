"""MISO's alignment losses and the hierarchical driver (reference: grid_opt/align/miso.py).

pairwise_loss_latent keeps the reference's value and pose gradients but not its waste:
grid features are read-only here (only submap poses are optimised), so they enter the
lookup detached and the dense, unused grad scatter of both submaps never runs; and the
L2 / L1 variants weight by the in-bound mask instead of compacting with nonzero, so an
iteration has no host sync."""
import logging

import numpy as np
import torch
import torch.nn.functional as F

import miso_amd.grid_opt.utils.utils as utils
import miso_amd.grid_opt.utils.utils_geometry as utils_geometry
from miso_amd.grid_opt.align.base import *   # noqa: F401,F403
from miso_amd.grid_opt.align.base import generic_align_multiple_submaps
from miso_amd.grid_opt.models.grid_atlas import GridAtlas
from miso_amd.grid_opt.utils.utils_geometry import transform_points_to

logger = logging.getLogger(__name__)


def _query_feature_readonly(submap, x, n_levels):
    """Features of levels 0..n_levels-1 only (the reference samples all levels and slices the
    channels, miso.py:187-188), with the grids detached."""
    from miso_amd import ops
    feats = [g.feature.detach() for g in submap.features[:n_levels]]
    meta = submap.features[0].grid_meta(submap.ignore_level_[:n_levels])
    return ops.encode(x, feats, meta)


def _src_features(grid_atlas, src_id, level, coords_from, n_levels):
    """Features of the source submap at its cached alignment coordinates.  They depend neither
    on the poses nor on the iteration, so they are computed once per (submap, level) and reused
    until a grid or the coordinate cache changes (the reference re-samples them every iteration)."""
    sub = grid_atlas.get_submap(src_id)
    key = (src_id, level, n_levels, coords_from.data_ptr(), coords_from.shape[0],
           tuple((g.feature.data_ptr(), g.feature._version) for g in sub.features[:n_levels]),
           tuple(bool(v) for v in sub.ignore_level_[:n_levels]))
    cache = grid_atlas.__dict__.setdefault('_align_src_cache', {})
    hit = cache.get((src_id, level))
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, _query_feature_readonly(sub, coords_from, n_levels))
        cache[(src_id, level)] = hit
    return hit[1]


def pairwise_loss_latent(grid_atlas: GridAtlas, data_loader, src_id: int, dst_id: int, level: int, fdim=4,
                         align_weight=3000, align_loss='L2', use_bound=True, stability_thresh=0,
                         covariance_thresh=None, subsample_points=None, trunc_factor=None, device="cuda:0"):
    """Residual between src's features at its cached voxel centres and dst's features at the
    same points mapped src -> world -> dst, over the channels of levels 0..level."""
    key = f'align_latent_level{level}_{src_id}_{dst_id}'
    assert src_id < grid_atlas.num_submaps and dst_id < grid_atlas.num_submaps
    if covariance_thresh is not None:
        raise NotImplementedError
    end_ch = fdim * (level + 1)
    sub_from, sub_to = grid_atlas.get_submap(src_id), grid_atlas.get_submap(dst_id)
    R_from, t_from = grid_atlas.updated_submap_pose(src_id, device)
    R_to, t_to = grid_atlas.updated_submap_pose(dst_id, device)
    coords_from = grid_atlas.coordinates_for_alignment(submap_id=src_id, level=level)
    if subsample_points is not None:
        k = min(subsample_points, coords_from.shape[0])
        coords_from = coords_from[np.random.choice(coords_from.shape[0], k, replace=False), :]
    fused_ok = (align_loss in ('L2', 'L1') and use_bound and stability_thresh <= 0 and trunc_factor is None
                and coords_from.is_cuda and fdim == sub_from.fdim)
    if fused_ok:
        # one kernel: both rigid maps, the bound mask, dst lookup, residual and the pose cotangents
        # (nothing below -- the (N,3) transforms, the mask -- is materialised on this path)
        from miso_amd import ops
        nlv = min(level + 1, sub_from.num_levels)
        f_from = (_src_features(grid_atlas, src_id, level, coords_from, nlv) if subsample_points is None
                  else _query_feature_readonly(sub_from, coords_from, nlv))
        feats_to = [g.feature.detach() for g in sub_to.features[:nlv]]
        meta_to = sub_to.features[0].grid_meta(sub_to.ignore_level_[:nlv])
        val = ops.pair_latent(R_from, t_from, R_to, t_to, coords_from, f_from, feats_to, meta_to, align_loss)
        return {key: val * align_weight}
    coords_world = utils_geometry.transform_points_to(coords_from, R_from, t_from)
    coords_to = utils_geometry.transfrom_points_from(coords_world, R_to, t_to)
    mask = torch.ones((coords_from.shape[0], 1), dtype=torch.bool, device=coords_from.device)
    if use_bound:
        mask = mask & utils_geometry.coords_in_bound(coords_to, sub_to.bound)
    if stability_thresh > 0:
        mu_to = sub_to.query_stability(coords_to)[:, 0:1]
        mu_from = sub_from.query_stability(coords_from)[:, 0:1]
        mask = mask & (mu_to > stability_thresh) & (mu_from > stability_thresh)
    if trunc_factor is not None:
        with torch.no_grad():
            near = torch.abs(sub_from(coords_from)) < trunc_factor * sub_from.cell_sizes[level]
        mask = mask & near
    if align_loss in ('L2', 'L1'):
        # sync-free: masked mean instead of nonzero-compaction (out-of-bound rows sample zeros)
        w = mask.to(coords_from.dtype)
        n_valid = w.sum().clamp(min=1.0)
        nlv = min(level + 1, sub_from.num_levels) if fdim == sub_from.fdim else sub_from.num_levels
        f_from = (_src_features(grid_atlas, src_id, level, coords_from, nlv) if subsample_points is None
                  else _query_feature_readonly(sub_from, coords_from, nlv))
        diff = (f_from[:, :end_ch] - _query_feature_readonly(sub_to, coords_to, nlv)[:, :end_ch]) * w
        if align_loss == 'L2':
            val = diff.pow(2).sum() / (n_valid * end_ch)
        else:
            val = torch.linalg.vector_norm(diff, dim=1).sum() / n_valid
        return {key: val * align_weight}
    if torch.count_nonzero(mask) == 0:
        return {key: torch.tensor(0)}
    rows = torch.nonzero(mask, as_tuple=False)[:, 0]
    nlv = sub_from.num_levels
    out_from = _query_feature_readonly(sub_from, coords_from[rows], nlv)[:, :end_ch]
    out_to = _query_feature_readonly(sub_to, coords_to[rows], nlv)[:, :end_ch]
    if align_loss == 'cos':
        val = torch.mean(1.0 - F.cosine_similarity(out_from, out_to, dim=1))
    elif align_loss == 'InfoNCE':
        val = utils.InfoNCE()(out_from, out_to)
    else:
        raise ValueError(f"Invalid align loss: {align_loss}!")
    return {key: val * align_weight}


def pairwise_loss_latent_batched(grid_atlas: GridAtlas, pairs, level: int, fdim=4, align_weight=3000,
                                 align_loss='L2', check_intersection=True, overlap_thresh=1e-2, device="cuda:0"):
    """pairwise_loss_latent for every (src, dst) of ``pairs`` in one autograd node (ops.pair_latent_multi):
    same values and pose gradients as the per-pair calls, including the overlap gate of
    generic_align_multiple_submaps (base.py:134; decided on the device), for the default options
    (L2 / L1, bound mask on, no stability / truncation pruning, no subsampling)."""
    import ctypes as C
    from miso_amd import ops
    assert align_loss in ('L2', 'L1')
    R_all, t_all = grid_atlas.updated_submap_poses_all(device)
    cache = grid_atlas.__dict__.setdefault('_align_grid_cache', {})
    plan = dict(pairs=list(pairs), loss_type=align_loss, coords=[], feats_src=[], grids=[], n_ch=[], keep=[],
                gate_pts=[] if check_intersection else None, overlap_thresh=float(overlap_thresh))
    for src_id, dst_id in pairs:
        sub_from, sub_to = grid_atlas.get_submap(src_id), grid_atlas.get_submap(dst_id)
        assert fdim == sub_from.fdim
        nlv = min(level + 1, sub_from.num_levels)
        coords = grid_atlas.coordinates_for_alignment(submap_id=src_id, level=level, brick_order=True)
        feats_to = [g.feature.detach() for g in sub_to.features[:nlv]]
        gkey = (dst_id, nlv, tuple(f.data_ptr() for f in feats_to), tuple(bool(v) for v in sub_to.ignore_level_[:nlv]))
        hit = cache.get((dst_id, nlv))
        if hit is None or hit[0] != gkey:
            meta_to = sub_to.features[0].grid_meta(sub_to.ignore_level_[:nlv])
            bmin = (C.c_float * 3)(*meta_to.bound_min)
            bmax = (C.c_float * 3)(*meta_to.bound_max)
            hit = (gkey, ops._fill_grid(feats_to, meta_to), bmin, bmax)
            cache[(dst_id, nlv)] = hit
        plan["coords"].append(coords)
        plan["feats_src"].append(ops._rows(_src_features(grid_atlas, src_id, level, coords, nlv)))
        plan["grids"].append(hit[1])
        plan["n_ch"].append(sum(int(f.shape[1]) for f in feats_to))
        plan["keep"].append(feats_to)
        if check_intersection:
            plan["gate_pts"].append((grid_atlas._finest_vertices(src_id), hit[2], hit[3]))
    # the index / count tensors of the plan depend only on the pair list: keep them across iterations
    ckey = (tuple(pairs), level, bool(check_intersection), tuple(plan["n_ch"]),
            tuple(c.shape[0] for c in plan["coords"]))
    cc = grid_atlas.__dict__.setdefault('_align_plan_const', {})
    if cc.get('key') == ckey:
        plan["_const"] = cc['const']
    losses = ops.pair_latent_multi(R_all, t_all, plan) * align_weight
    cc['key'], cc['const'] = ckey, plan.get("_const")
    return {f'align_latent_level{level}_{a}_{b}': losses[i] for i, (a, b) in enumerate(pairs)}


def latent_pair_inputs(grid_atlas: GridAtlas, pairs, level: int, fdim=4, check_intersection=True):
    """Per-pair inputs of ops.AlignPlan for pairwise_loss_latent at ``level`` (default options: L2 / L1, bound mask
    on, no stability / truncation pruning, no subsampling): the source's cached alignment vertices and its features
    there (pose independent: computed once per (submap, level)), the destination's levels 0..level, and the
    source's finest-level vertices for the overlap gate of base.py:134."""
    out = []
    for src_id, dst_id in pairs:
        sub_from, sub_to = grid_atlas.get_submap(src_id), grid_atlas.get_submap(dst_id)
        assert fdim == sub_from.fdim
        nlv = min(level + 1, sub_from.num_levels)
        coords = grid_atlas.coordinates_for_alignment(submap_id=src_id, level=level, brick_order=True)
        out.append(dict(src=src_id, dst=dst_id, coords=coords,
                        feats_src=_src_features(grid_atlas, src_id, level, coords, nlv),
                        feats_dst=[g.feature.detach() for g in sub_to.features[:nlv]],
                        meta_dst=sub_to.features[0].grid_meta(sub_to.ignore_level_[:nlv]),
                        gate_pts=grid_atlas._finest_vertices(src_id) if check_intersection else None,
                        gate_dims=tuple(int(v) for v in reversed(sub_from.features[-1].feature.shape[2:]))))
    return out


def pairwise_loss_sdf(grid_atlas: GridAtlas, data_loader, src_id: int, dst_id: int, align_weight=3000,
                      align_loss='L2', use_bound=True, stability_thresh=0, covariance_thresh=None,
                      subsample_points=None, gm_scale_sdf=0.1, device="cuda:0"):
    """SDF-space variant on the observed samples of src's keyframes (reference :14-113)."""
    from miso_amd.grid_opt.loss import transform_by_keyframe
    assert src_id < grid_atlas.num_submaps and dst_id < grid_atlas.num_submaps
    if covariance_thresh is not None:
        raise NotImplementedError
    model_input, gt = utils.get_batch(data_loader, device)
    sub_from, sub_to = grid_atlas.get_submap(src_id), grid_atlas.get_submap(dst_id)
    owner = grid_atlas.submap_id_for_kf_batch(kf_ids=model_input['sample_frame_ids'][0, :, 0])
    rows = torch.nonzero(owner == src_id, as_tuple=False).squeeze(1)
    if rows.numel() == 0:
        return {}
    coords_kf = model_input['coords_frame'][0, rows, :]
    kf_idxs = model_input['sample_frame_ids'][0, rows, 0]
    coords_from = transform_by_keyframe(coords_kf, kf_idxs,
                                        lambda k: grid_atlas.updated_kf_pose_in_submap(k, src_id))
    mask = gt['sdf_valid'][0][rows, :] == 1
    R_from, t_from = grid_atlas.updated_submap_pose(src_id, device)
    R_to, t_to = grid_atlas.updated_submap_pose(dst_id, device)
    coords_world = utils_geometry.transform_points_to(coords_from, R_from, t_from)
    coords_to = utils_geometry.transfrom_points_from(coords_world, R_to, t_to)
    if subsample_points is not None:
        k = min(subsample_points, coords_from.shape[0])
        pick = np.random.choice(coords_from.shape[0], k, replace=False)
        coords_from, coords_to, mask = coords_from[pick], coords_to[pick], mask[pick]
    if use_bound:
        mask = mask & utils_geometry.coords_in_bound(coords_to, sub_to.bound)
    if stability_thresh > 0:
        mask = mask & (sub_to.query_stability(coords_to)[:, 0:1] > stability_thresh) \
            & (sub_from.query_stability(coords_from)[:, 0:1] > stability_thresh)
    keep = torch.nonzero(mask, as_tuple=False)[:, 0]
    resid = sub_from(coords_from[keep]) - sub_to(coords_to[keep])
    key = f'align_sdf_{src_id}_{dst_id}'
    if align_loss == 'L2':
        val = torch.mean(resid ** 2)
    elif align_loss == 'L1':
        val = torch.mean(torch.linalg.vector_norm(resid, dim=1))
    elif align_loss == 'GM':
        e = resid.detach()
        val = torch.mean(gm_scale_sdf / (gm_scale_sdf + e ** 2) ** 2 * resid ** 2)
    else:
        raise ValueError(f"Invalid align loss: {align_loss}!")
    return {key: val * align_weight}


def latent_loss_for_level(grid_atlas: GridAtlas, level: int, align_weight=3000, align_loss="L2", use_bound=True,
                          stability_thresh=0, subsample_points=None, device="cuda:0"):
    """The pair-loss callable generic_align_multiple_submaps takes (``loss(atlas, loader, src, dst) -> dict``) for
    pairwise_loss_latent at ``level``, carrying what lets the driver avoid the per-pair host work when the default
    options are on (L2 / L1, bound mask, no stability pruning, no subsampling) and the atlas lives on the GPU:
    ``device_gate`` (overlap decided on the device), ``batched`` (all pairs behind one autograd node) and ``fused``
    (the whole pose-Adam loop on the device, base.fused_alignment_loop)."""
    from miso_amd import ops as _ops

    def latent(atlas, loader, a, b):
        return pairwise_loss_latent(atlas, loader, a, b, level=level, align_weight=align_weight,
                                    align_loss=align_loss, use_bound=use_bound, stability_thresh=stability_thresh,
                                    subsample_points=subsample_points, device=device)
    default_opts = (align_loss in ('L2', 'L1') and use_bound and stability_thresh <= 0 and subsample_points is None)
    on_gpu = (default_opts and str(device).startswith('cuda')
              and grid_atlas.get_submap(0).features[0].feature.is_cuda)
    # the fused pair kernel is cheap enough to run on non-overlapping pairs too: let the driver gate it on the
    # device instead of synchronising on check_submap_intersection per pair
    latent.device_gate = on_gpu
    if on_gpu:
        def latent_all(atlas, pairs, check_intersection):
            return pairwise_loss_latent_batched(atlas, pairs, level=level, fdim=atlas.get_submap(0).fdim,
                                                align_weight=align_weight, align_loss=align_loss,
                                                check_intersection=check_intersection, device=device)
        latent.batched = latent_all
    if on_gpu or (default_opts and getattr(_ops.AlignPlan, 'cpu_ok', False)):
        latent.fused = dict(align_loss=align_loss, align_weight=align_weight, level=level,
                            inputs=lambda atlas, pairs, chk: latent_pair_inputs(
                                atlas, pairs, level=level, fdim=atlas.get_submap(0).fdim, check_intersection=chk))
    return latent


def align_multiple_submaps_hierarchical(grid_atlas: GridAtlas, dataset, level_iters=10, finetune_iters=10,
                                        level_thresh=0.0, lr=1e-2, align_weight=3000, align_loss="L2",
                                        use_bound=True, stability_thresh=0, subsample_points=None,
                                        latent_levels=None, skip_finetune=False, submap_pairs=None,
                                        pose_reg_weight=0, pose_thresh_m=1.0, pose_thresh_rad=1.0,
                                        gm_scale_sdf=0.1, device="cuda:0", verbose=True, save_iterations=False):
    """Coarse-to-fine alignment in latent space, optional SDF-space fine-tune (reference :217-322).
    ``info`` keys (hier_latent_level{l}_{loss}, hier_sdf_{loss}, cpu/gpu_time_sec) as upstream."""
    grid_atlas.precompute_coordinates_for_alignment()
    info = dict()
    cpu_total = gpu_total = 0
    common = dict(lr=lr, submap_pairs=submap_pairs, pose_reg_weight=pose_reg_weight, pose_thresh_m=pose_thresh_m,
                  pose_thresh_rad=pose_thresh_rad, verbose=verbose, save_iterations=save_iterations)
    levels = range(grid_atlas.num_levels) if latent_levels is None else latent_levels
    for lvl in levels:
        latent = latent_loss_for_level(grid_atlas, lvl, align_weight=align_weight, align_loss=align_loss,
                                       use_bound=use_bound, stability_thresh=stability_thresh,
                                       subsample_points=subsample_points, device=device)
        name = f'hier_latent_level{lvl}_{align_loss}'
        res = generic_align_multiple_submaps(grid_atlas, dataset, (name, latent), num_iters=level_iters,
                                             rel_change_thresh=level_thresh, **common)
        cpu_total += res['cpu_time_sec']
        gpu_total += res['gpu_time_sec']
        info[name] = res
    if not skip_finetune:
        sdf_loss_type = 'L2' if align_loss == 'cos' else align_loss

        def sdf(atlas, loader, a, b):
            return pairwise_loss_sdf(atlas, loader, a, b, align_weight=align_weight, align_loss=sdf_loss_type,
                                     use_bound=use_bound, stability_thresh=stability_thresh,
                                     subsample_points=subsample_points, gm_scale_sdf=gm_scale_sdf, device=device)
        name = f'hier_sdf_{sdf_loss_type}'
        res = generic_align_multiple_submaps(grid_atlas, dataset, (name, sdf), num_iters=finetune_iters, **common)
        cpu_total += res['cpu_time_sec']
        gpu_total += res['gpu_time_sec']
        info[name] = res
    info['cpu_time_sec'] = cpu_total
    info['gpu_time_sec'] = gpu_total
    return info
