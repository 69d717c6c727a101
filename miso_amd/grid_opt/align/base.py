"""Pose-only Adam loops over submap pairs (reference: grid_opt/align/base.py)."""
import logging

import torch
from torch.utils.data import DataLoader, Dataset
from miso_amd.grid_opt.utils.utils import collate_batch_of_one

import miso_amd.grid_opt.utils.utils as utils
import miso_amd.grid_opt.utils.utils_geometry as utils_geometry
from miso_amd.grid_opt.models.grid_atlas import GridAtlas
from miso_amd.optim import DenseAdam

logger = logging.getLogger(__name__)


def grid_atlas_pose_l2_loss(model: GridAtlas, weight=1e3):
    out = {}
    for s in range(model.num_submaps):
        out[f'submap{s}_l2_reg_R'] = weight * torch.sum(model.rotation_corrections[s] ** 2)
        out[f'submap{s}_l2_reg_t'] = weight * torch.sum(model.translation_corrections[s] ** 2)
    return out


def grid_atlas_pose_trust_region_loss(model: GridAtlas, thresh_rad, thresh_m, weight=1e3):
    relu = torch.nn.functional.relu
    out = {}
    for s in range(model.num_submaps):
        out[f'submap{s}_trust_region_R'] = weight * relu(torch.linalg.norm(model.rotation_corrections[s]) - thresh_rad)
        out[f'submap{s}_trust_region_t'] = weight * relu(torch.linalg.norm(model.translation_corrections[s]) - thresh_m)
    return out


def iteration_results_helper(model: GridAtlas):
    """(S,4,4) snapshot of the current submap poses."""
    assert isinstance(model, GridAtlas), "Model must be an instance of GridAtlas."
    poses = torch.zeros((model.num_submaps, 4, 4), dtype=torch.float32, device=model.device)
    for s in range(model.num_submaps):
        poses[s] = utils_geometry.pose_matrix(*model.updated_submap_pose(s))
    return poses.detach()


def _adam_step(optimizer, loss_dict, what):
    total = sum(loss_dict.values())
    if not torch.isnan(total):
        total.backward(retain_graph=False)
        optimizer.step()
    else:
        logger.warning(f"Loss at {what} is nan! Skip backward step.")
    return total


def generic_align_submap_pair(grid_atlas: GridAtlas, dataset: Dataset, src_id: int, dst_id: int,
                              pairwise_loss_tuple, num_iters=10, lr=1e-2, rel_change_thresh=0, verbose=True):
    """Align dst to src by optimising dst's pose only (reference :41-87)."""
    assert src_id < grid_atlas.num_submaps and dst_id < grid_atlas.num_submaps
    grid_dst = grid_atlas.get_submap(dst_id)
    loader = DataLoader(dataset, shuffle=True, batch_size=1, num_workers=0, collate_fn=collate_batch_of_one)
    optimizer = DenseAdam([{'params': grid_atlas.params_for_submap_pose(dst_id), 'lr': lr}], lr=lr)
    loss_name, loss_func = pairwise_loss_tuple
    prev = None
    timer = utils.PerfTimer(activate=True)
    it = 0
    while it <= num_iters:      # upstream runs num_iters + 1 iterations
        optimizer.zero_grad()
        total = _adam_step(optimizer, loss_func(grid_atlas, loader, src_id, dst_id), f"iter {it}")
        cur = [p.clone().detach() for p in grid_dst.params_for_poses()]
        change = utils.relative_param_change(cur, prev)
        prev = cur
        if verbose:
            logger.info(f"AlignPair_{loss_name} iteration {it}: loss = {total.item():.2e}, pose_relchange={change:.2e}")
        if change < rel_change_thresh:
            break
        it += 1
    cpu_time, gpu_time = timer.check()
    return {'cpu_time_sec': cpu_time, 'gpu_time_sec': gpu_time}


def _captured_alignment_loop(grid_atlas, params, batched, submap_pairs, check_intersection, reg, lr, num_iters,
                             loss_name, verbose):
    """The whole pose-Adam iteration -- zero the gradients, all pair losses behind one autograd node, backward
    through the batched exponential map, Adam on the 6(S-1) pose numbers -- captured ONCE in a HIP graph and
    replayed: an iteration is ~60 tiny launches plus the pair kernels, and eagerly the host needs 3x as long to
    issue them as the GPU to run them.  Adam is torch's own with ``capturable=True`` (step counts on the device;
    same formula as the reference's torch.optim.Adam).  The first three iterations run eagerly on a side stream,
    as graph capture wants; losses and relative pose changes of every iteration are kept on the device and
    logged afterwards.  Returns False (nothing done) if the capture cannot be made; the caller's eager loop runs."""
    dev = params[0].device
    total_iters = num_iters + 1
    if total_iters < 8:
        return False
    hist = torch.zeros((total_iters, 2), device=dev)              # loss, relative change
    it_dev = torch.zeros(1, device=dev, dtype=torch.long)
    prev = [torch.zeros_like(p) for p in params]
    one = torch.ones(1, device=dev, dtype=torch.long)
    try:        # one multi-tensor kernel for the 2(S-1) pose tensors instead of ~15 launches per tensor
        optimizer = torch.optim.Adam(params, lr=lr, capturable=True, fused=True)
    except (RuntimeError, TypeError, ValueError):
        optimizer = torch.optim.Adam(params, lr=lr, capturable=True, foreach=False)

    def iteration():
        loss_dict = dict(batched(grid_atlas, submap_pairs, check_intersection))
        if reg is not None:
            loss_dict.update(reg())
        total = sum(loss_dict.values())
        total.backward()
        if not verbose:                      # nothing to report afterwards: no bookkeeping kernels in the graph
            optimizer.step()
            return
        with torch.no_grad():
            for q, p in zip(prev, params):
                q.copy_(p)
        optimizer.step()
        with torch.no_grad():
            num = sum(torch.sum((p - q) ** 2) for p, q in zip(params, prev))
            den = sum(torch.sum(q ** 2) for q in prev)
            # upstream compares the parameters AFTER iterations k and k-1 (base.py:152-158)
            row = torch.stack((total.detach(), torch.sqrt(num / den))).reshape(1, 2)
            hist.index_copy_(0, it_dev, row)
            it_dev.add_(one)

    snapshot = [p.detach().clone() for p in params]
    graph = None
    try:
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(3):
                optimizer.zero_grad(set_to_none=True)
                iteration()
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            iteration()
    except Exception as exc:                                        # pragma: no cover - depends on the runtime
        logger.warning(f"alignment loop not captured ({type(exc).__name__}: {exc}); running it op by op")
        with torch.no_grad():
            for p, q in zip(params, snapshot):
                p.copy_(q)
                p.grad = None
        return False
    for _ in range(total_iters - 3):
        graph.replay()
    for p in params:
        p.grad = None
    if verbose:
        rows = hist.cpu().tolist()
        for it, (loss, change) in enumerate(rows):
            # the change upstream logs at iteration k is between the parameters after k and after k-1: inf at k = 0
            shown = float('inf') if it == 0 else rows[it][1]
            logger.info(f"AlignMulti_{loss_name} iteration {it}: loss = {loss:.2e}, pose_relchange={shown:.2e}, "
                        f"lr={lr:.2e}")
    return True


def generic_align_multiple_submaps(grid_atlas: GridAtlas, dataset: Dataset, pairwise_loss_tuple, num_iters=10,
                                   lr=1e-2, rel_change_thresh=0, submap_pairs=None, check_intersection=True,
                                   pose_reg_weight=0, pose_thresh_rad=1.0, pose_thresh_m=1.0, verbose=True,
                                   save_iterations=False):
    """Adam over the pose corrections of submaps 1..S-1 (submap 0 stays fixed) on the sum of
    pairwise losses (reference :89-163)."""
    def pose_params():
        return [p for s in range(1, grid_atlas.num_submaps) for p in grid_atlas.params_for_submap_pose(s)]

    optimizer = DenseAdam([{'params': pose_params(), 'lr': lr}], lr=lr)
    loader = DataLoader(dataset, shuffle=True, batch_size=1, num_workers=0, collate_fn=collate_batch_of_one)
    loss_name, loss_func = pairwise_loss_tuple
    if submap_pairs is None:
        n = grid_atlas.num_submaps
        submap_pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    timer = utils.PerfTimer(activate=True)
    iteration_results = dict()
    prev = None
    it = 0
    batched_fn = getattr(loss_func, 'batched', None)
    params = pose_params()
    if (batched_fn is not None and not save_iterations and rel_change_thresh <= 0 and params
            and all(p.is_cuda and p.requires_grad for p in params)
            and not getattr(grid_atlas, 'no_captured_alignment', False)):
        reg = None
        if pose_reg_weight > 0:
            reg = lambda: grid_atlas_pose_trust_region_loss(grid_atlas, thresh_rad=pose_thresh_rad,        # noqa: E731
                                                            thresh_m=pose_thresh_m, weight=pose_reg_weight)
        if _captured_alignment_loop(grid_atlas, params, batched_fn, submap_pairs, check_intersection, reg, lr,
                                    num_iters, loss_name, verbose):
            it = num_iters + 1
    while it <= num_iters:
        if save_iterations:
            iteration_results[it] = iteration_results_helper(grid_atlas)
        optimizer.zero_grad()
        loss_dict = {}
        # one backward per iteration over the summed pair losses: the updated submap poses (and the
        # graph through so3_exp_map) are shared by all pairs
        batched = getattr(loss_func, 'batched', None)
        if batched is not None:
            loss_dict.update(batched(grid_atlas, submap_pairs, check_intersection))
        with grid_atlas.pose_cache():
            for src_id, dst_id in (submap_pairs if batched is None else ()):
                gate = None
                if check_intersection:
                    inter = grid_atlas.check_submap_intersection(src_id, dst_id)
                    if getattr(loss_func, 'device_gate', False) and isinstance(inter, torch.Tensor) and inter.is_cuda:
                        # cheap fused pair loss: evaluate it regardless and multiply by the 0/1 overlap
                        # flag on the device instead of stalling the host on it for every pair
                        gate = inter.to(torch.float32)
                    elif not bool(inter):
                        continue
                pair = loss_func(grid_atlas, loader, src_id, dst_id)
                loss_dict.update({k: torch.nan_to_num(v) * gate if gate is not None else torch.nan_to_num(v)
                                  for k, v in pair.items()})
        if pose_reg_weight > 0:
            loss_dict.update(grid_atlas_pose_trust_region_loss(grid_atlas, thresh_rad=pose_thresh_rad,
                                                               thresh_m=pose_thresh_m, weight=pose_reg_weight))
        total = _adam_step(optimizer, loss_dict, f"iter {it}")
        cur = [p.clone().detach() for p in pose_params()]
        change = utils.relative_param_change(cur, prev)
        prev = cur
        if verbose:
            logger.info(f"AlignMulti_{loss_name} iteration {it}: loss = {float(total):.2e}, "
                        f"pose_relchange={change:.2e}, lr={lr:.2e}")
        if change < rel_change_thresh:
            break
        it += 1
    cpu_time, gpu_time = timer.check()
    return {'cpu_time_sec': cpu_time, 'gpu_time_sec': gpu_time, 'iteration_results': iteration_results}
