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


def _adam_step_reduced(optimizer, params, loss_dict, reduce, what):
    """_adam_step with the pair list sharded over ranks: backward of THIS rank's losses, the pose gradients and the loss
    packed into one flat tensor, ``reduce`` (an all-reduce SUM) over the ranks, then the reference's NaN guard
    (base.py:147-151) and the optimizer step on the summed gradient -- the same on every rank.  A rank without a loss
    of its own (all its pairs gated out) contributes zeros; a NaN anywhere reaches every rank through the sum."""
    dev = params[0].device
    total = sum(loss_dict.values()) if loss_dict else torch.zeros((), device=dev)
    total = total if torch.is_tensor(total) else torch.tensor(float(total), device=dev)
    if total.requires_grad and not bool(torch.isnan(total)):
        total.backward(retain_graph=False)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]
                     + [total.detach().reshape(1).to(params[0].dtype)])
    if reduce is not None:
        reduce(flat)
    total_all = flat[-1]
    if bool(torch.isnan(total_all)) or bool(torch.isnan(flat[:-1]).any()):
        logger.warning(f"Loss at {what} is nan! Skip backward step.")
        return total_all
    o = 0
    for p in params:
        n = p.numel()
        p.grad = flat[o:o + n].reshape(p.shape).clone()
        o += n
    optimizer.step()
    return total_all


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


def fused_alignment_loop(grid_atlas, fused, submap_pairs, check_intersection, lr, num_iters, rel_change_thresh,
                         pose_reg_weight, pose_thresh_rad, pose_thresh_m, verbose, save_iterations, loss_name,
                         reduce=None, my_pairs=None, use_graph=True):
    """The whole loop of generic_align_multiple_submaps on the device (ops.AlignPlan: miso_align_iteration_a / _b).

    One iteration is five launches whatever the number of pairs -- poses from the corrections, the overlap gate
    and the latent residual of every pair (one launch each, grid.y = pair), the pull-back of the pose cotangents
    through R0 Exp(dr), then regulariser + NaN guard + Adam + early-stop test -- and the host reads nothing until
    the loop ends: per-iteration losses, relative changes and (save_iterations) the (S,4,4) pose snapshots of
    iteration_results_helper are written to a device ring by the kernels.  So the loop also stays on the device
    under the reference's own alignment config (verbose: True, save_iterations: True, configs/rgbd/scannet.yaml:65-66).

    fused: dict(inputs=callable(atlas, pairs, check_intersection) -> list of pair dicts for ops.AlignPlan,
    align_loss, align_weight[, overlap_thresh]).  reduce / my_pairs: the multi-rank hook of miso_amd.dist -- this
    rank evaluates ``my_pairs`` only and ``reduce(flat)`` sums the 6S + 1 floats over ranks between the two halves
    of the iteration.  Returns the ``iteration_results`` dict."""
    from miso_amd import ops
    S = grid_atlas.num_submaps
    dev = grid_atlas.rotation_corrections[0].device
    pairs = list(submap_pairs if my_pairs is None else my_pairs)
    total_iters = num_iters + 1                       # upstream runs num_iters + 1 iterations (base.py:127)
    R0 = torch.stack([R.to(dev) for R in grid_atlas.R_world_submap_list])
    t0 = torch.stack([t.to(dev) for t in grid_atlas.t_world_submap_list])
    plan = ops.AlignPlan(R0, t0, fused['inputs'](grid_atlas, pairs, check_intersection),
                         loss_type=fused['align_loss'], align_weight=fused['align_weight'],
                         overlap_thresh=fused.get('overlap_thresh', 1e-2), lr=lr,
                         reg_weight=pose_reg_weight if pose_reg_weight > 0 else 0.0, reg_thresh_rad=pose_thresh_rad,
                         reg_thresh_m=pose_thresh_m, rel_change_thresh=rel_change_thresh, ring_iters=total_iters,
                         save_poses=save_iterations)
    with torch.no_grad():
        dr = torch.cat([p.detach().reshape(1, 3) for p in grid_atlas.rotation_corrections], dim=0)
        dt = torch.cat([p.detach().reshape(1, 3) for p in grid_atlas.translation_corrections], dim=0)
        plan.params.copy_(torch.cat((dr, dt), dim=1))

    def iteration():
        plan.iteration_a()
        if reduce is not None:
            reduce(plan.flat_reduce)
        plan.iteration_b()

    done = 0
    graph = graph_many = None
    UNROLL = 8
    if use_graph and dev.type == 'cuda' and total_iters >= 8:
        # level 0 iterations are ~30 us of GPU work: replay them instead of issuing five launches each.  With the
        # multi-rank hook the iteration is TWO graphs around the collective -- half A (poses, gates, pair stage, pull-back),
        # all-reduce of `flat` issued on the same stream, half B (regulariser, guard, Adam) -- and still no host sync
        # (round 2 dropped the capture altogether under `reduce`: five eager launches + an all-reduce for ~10 us of
        # work per rank at 8 ranks).
        try:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                iteration()                                           # iteration 0, eagerly (capture wants a warm-up)
            cur.wait_stream(side)
            done = 1
            def capture(fn):
                # capture_begin / capture_end on the side stream, NOT the torch.cuda.graph context manager: that one
                # runs gc.collect() and torch.cuda.empty_cache() on entry, and a loop is captured per level per call
                # (emptying the allocator's cache makes every allocation of the next set-up a hipMalloc)
                g = torch.cuda.CUDAGraph()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    g.capture_begin(capture_error_mode="thread_local")
                    try:
                        fn()
                    finally:
                        g.capture_end()
                torch.cuda.current_stream().wait_stream(side)
                return g
            graph = capture(iteration) if reduce is None else (capture(plan.iteration_a), capture(plan.iteration_b))
            if reduce is None and total_iters - done >= 2 * UNROLL:
                # a replay follows the previous one after ~6 us of device idle time, 5-8 % of a level-0
                # iteration: UNROLL iterations per replay while that many are left (they are identical launches -- the
                # iteration counter, the ring slot and the early-stop flag live on the device)
                def several():
                    for _ in range(UNROLL):
                        iteration()
                graph_many = capture(several)
        except Exception as exc:                                    # pragma: no cover - depends on the runtime
            logger.warning(f"alignment iteration not captured ({type(exc).__name__}: {exc}); issuing it launch by launch")
            graph = graph_many = None
    check_every = 16 if rel_change_thresh > 0 else 0
    import time
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    t_loop, n_loop = time.perf_counter(), done
    while done < total_iters:
        if isinstance(graph, tuple):
            graph[0].replay()
            reduce(plan.flat_reduce)
            graph[1].replay()
        elif graph_many is not None and total_iters - done >= UNROLL and (not check_every or done % UNROLL == 0):
            graph_many.replay()
            done += UNROLL - 1
        elif graph is not None:
            graph.replay()
        else:
            iteration()
        done += 1
        if check_every and done % check_every == 0 and plan.ctrl()['stopped']:
            break                                                    # further iterations would change nothing
    ctrl = plan.ctrl()                                                # (one host read: the loop is over)
    # wall time of the iterations proper (plan set-up, warm-up and capture excluded), for whoever times the loop
    grid_atlas.__dict__['_last_align_loop'] = dict(seconds=time.perf_counter() - t_loop, iterations=done - n_loop)
    with torch.no_grad():
        prm = plan.params.clone()
        for s in range(1, S):                                        # submap 0 is fixed
            grid_atlas.rotation_corrections[s].copy_(prm[s, :3].reshape(1, 3))
            grid_atlas.translation_corrections[s].copy_(prm[s, 3:].reshape(3, 1))
    ran = ctrl['iterations']
    iteration_results = dict()
    if verbose or save_iterations or ctrl['skipped']:
        ring = plan.ring()[:ran].cpu()
        for it in range(ran):
            if save_iterations:
                iteration_results[it] = ring[it, 2:].reshape(S, 4, 4).to(dev)
            if verbose:
                logger.info(f"AlignMulti_{loss_name} iteration {it}: loss = {ring[it, 0].item():.2e}, "
                            f"pose_relchange={ring[it, 1].item():.2e}, lr={lr:.2e}")
        if ctrl['skipped']:
            logger.warning(f"AlignMulti_{loss_name}: loss was nan in {ctrl['skipped']} iteration(s); their steps were skipped.")
    return iteration_results


def generic_align_multiple_submaps(grid_atlas: GridAtlas, dataset: Dataset, pairwise_loss_tuple, num_iters=10,
                                   lr=1e-2, rel_change_thresh=0, submap_pairs=None, check_intersection=True,
                                   pose_reg_weight=0, pose_thresh_rad=1.0, pose_thresh_m=1.0, verbose=True,
                                   save_iterations=False, my_pairs=None, reduce=None):
    """Adam over the pose corrections of submaps 1..S-1 (submap 0 stays fixed) on the sum of
    pairwise losses (reference :89-163).

    my_pairs / reduce: the multi-rank hook of miso_amd.dist for pair losses WITHOUT a fused plan (the SDF fine-tune
    stage, align/miso.py pairwise_loss_sdf; reference miso.py:283-319): this rank evaluates ``my_pairs`` only -- the
    trust-region term on the rank that holds the first pair of the list, so that it is counted once -- and
    ``reduce(flat)`` sums the pose gradients and the loss over the ranks before the NaN guard and the identical Adam
    step on every rank (replicas stay equal; results equal the single-process loop up to the order of the sum)."""
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
    params = pose_params()
    fused = getattr(loss_func, 'fused', None)
    sharded = reduce is not None or my_pairs is not None
    local_pairs = submap_pairs if my_pairs is None else list(my_pairs)
    adds_reg = my_pairs is None or (len(submap_pairs) > 0 and submap_pairs[0] in local_pairs) or \
        (len(submap_pairs) == 0)
    if (fused is not None and params and all(p.requires_grad for p in params) and not sharded
            and not getattr(grid_atlas, 'no_fused_alignment', False)):
        iteration_results = fused_alignment_loop(grid_atlas, fused, submap_pairs, check_intersection, lr, num_iters,
                                                 rel_change_thresh, pose_reg_weight, pose_thresh_rad, pose_thresh_m,
                                                 verbose, save_iterations, loss_name)
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
            loss_dict.update(batched(grid_atlas, local_pairs, check_intersection))
        with grid_atlas.pose_cache():
            for src_id, dst_id in (local_pairs if batched is None else ()):
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
        if pose_reg_weight > 0 and adds_reg:
            loss_dict.update(grid_atlas_pose_trust_region_loss(grid_atlas, thresh_rad=pose_thresh_rad,
                                                               thresh_m=pose_thresh_m, weight=pose_reg_weight))
        if sharded:
            total = _adam_step_reduced(optimizer, params, loss_dict, reduce, f"iter {it}")
        else:
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
