"""Submap-parallel execution over ``torch.distributed`` -- one process per GPU, backend
``nccl`` (= RCCL over xGMI on MI355X), ``gloo`` in CPU tests.

The reference is single-process (no distributed code at all); the shard
unit follows from its data model: every submap is an independent GridNet with its own
grids, keyframe poses and optimiser state (grid_opt/models/grid_atlas.py:146-150).

* mapping   rank r owns submaps {s : s % world == r}; NO per-step collective (decoder
            frozen, grids disjoint).  ``sync_submaps`` afterwards: every owner packs ALL it
            owns into one flat buffer and broadcasts it -- `world` collectives of ~80 MB
            (61 MiB of grids per ScanNet submap) instead of one per parameter and buffer
            (~10 per submap, most of them a few bytes: latency, not bytes).
* alignment grids are read-only and replicated; the pair list is dealt by estimated cost
            (source vertices + in-bound vertices at the start poses, longest first onto the
            least loaded rank: a gated pair costs a fraction of an overlapping one); each
            iteration ends with ONE all-reduce(SUM) of a flat fp32 buffer holding the pose
            gradients and the loss (7S+1 floats: latency bound, xGMI bandwidth irrelevant)
            between two captured halves, then the identical Adam step everywhere.  A level whose whole
            iteration is cheaper than what sharding adds (the all-reduce + a second graph replay; cfg-4's
            level 0 is 82 us on ONE GPU) runs REPLICATED instead: every rank all pairs, no collective,
            bit-identical by construction (``alignment_mode``).  The deal itself is rank 0's, broadcast.
"""
from __future__ import annotations

import logging
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

import miso_amd.grid_opt.utils.utils as utils

logger = logging.getLogger(__name__)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun
    contract).  Returns (rank, world).  A single process needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def owned_submaps(num_submaps: int, rank: Optional[int] = None, world: Optional[int] = None) -> List[int]:
    r, w = rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    return [s for s in range(num_submaps) if s % world == rank]


def owner_of(submap_id: int, world: Optional[int] = None) -> int:
    return submap_id % (rank_world()[1] if world is None else world)


def _needs_host_staging(t: torch.Tensor) -> bool:
    """gloo moves host memory: device tensors are staged through the host (development runs that put several ranks
    on one GPU; production is nccl = RCCL, which takes the device pointer as is)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_sum(t: torch.Tensor, always: bool = False) -> torch.Tensor:
    """In-place SUM over ranks of a small flat tensor (the 7S + 1 floats of an alignment iteration).
    always: issue the collective even in a group of one rank (the single-GPU RCCL smoke test)."""
    if rank_world()[1] == 1 and not (always and dist.is_available() and dist.is_initialized()):
        return t
    if _needs_host_staging(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def _broadcast(t: torch.Tensor, src: int):
    if _needs_host_staging(t):
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def _flat_view(t: torch.Tensor) -> torch.Tensor:
    """Dense 1-D view of a parameter in its physical order (channels-last grids included)."""
    if t.ndim == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
        return t.permute(0, 2, 3, 4, 1).reshape(-1)
    return t.contiguous().view(-1) if not t.is_contiguous() else t.view(-1)


def map_submaps_parallel(atlas, map_one: Callable[[int], None], sync: bool = True):
    """Run ``map_one(submap_id)`` for the submaps this rank owns (build_submaps.py:133-134
    runs them one after the other on one GPU); optionally broadcast the results."""
    for s in owned_submaps(atlas.num_submaps):
        map_one(s)
    if sync:
        sync_submaps(atlas)


def _pack_list(atlas, submaps):
    """The tensors (parameters, then buffers, in module order) of `submaps`, and their byte sizes padded to 16."""
    ts = [t.data for s in submaps for t in list(atlas.get_submap(s).parameters()) + list(atlas.get_submap(s).buffers())]
    return ts, [(t.numel() * t.element_size() + 15) // 16 * 16 for t in ts]


@torch.no_grad()
def sync_submaps(atlas, always: bool = False):
    """Every rank ends up with every submap's parameters and buffers.  One collective per OWNER: rank r packs all the
    tensors of the submaps it owns (physical order: channels-last grids stay channels-last) into one byte buffer,
    broadcasts it, and everybody else unpacks.  The broadcasts of all owners are issued back to back and waited for
    together.  (Round 2 sent one blocking broadcast per parameter and buffer -- ~80 collectives for 8 submaps, most of
    them a handful of bytes.)  always: also in a group of one rank (RCCL smoke test)."""
    _, world = rank_world()
    if world == 1 and not (always and dist.is_available() and dist.is_initialized()):
        return
    rank = rank_world()[0]
    S = atlas.num_submaps
    dev = next(atlas.get_submap(0).parameters()).device
    bufs, works = [], []
    for owner in range(world):
        subs = owned_submaps(S, owner, world)
        ts, sizes = _pack_list(atlas, subs)
        if not ts:
            bufs.append(None)
            continue
        buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        if owner == rank:
            off = 0
            for t, nb in zip(ts, sizes):
                src = _flat_view(t)
                buf[off:off + src.numel() * src.element_size()].copy_(src.view(torch.uint8))
                off += nb
        bufs.append((buf, ts, sizes))
        if _needs_host_staging(buf):
            h = buf.cpu()
            dist.broadcast(h, src=owner)
            buf.copy_(h)
        else:
            works.append(dist.broadcast(buf, src=owner, async_op=True))
    for w in works:
        w.wait()
    for owner, item in enumerate(bufs):
        if item is None or owner == rank:
            continue
        buf, ts, sizes = item
        off = 0
        for t, nb in zip(ts, sizes):
            n = t.numel() * t.element_size()
            dst = _flat_view(t)
            src = buf[off:off + n].view(t.dtype)
            if dst.data_ptr() == t.data_ptr():
                dst.copy_(src)
            else:                                   # a strided parameter: _flat_view made a copy, write it back
                t.copy_(src.view_as(t))
            off += nb


def pair_costs(grid_atlas, pairs) -> List[float]:
    """Relative cost of every pair for the fused pair kernel at the CURRENT poses: every source vertex is transformed
    and tested (12 B), an in-bound one gathers 8 corners per level and ~300 flops on top.  Estimated on the coarsest
    level's vertex lattice with plain tensor ops (identical on every rank: same inputs, same ops), one read-back."""
    if not pairs:
        return []
    fracs = []
    with torch.no_grad():
        world_pts, inv = {}, {}
        for a, b in pairs:
            if a not in world_pts:                      # once per source submap, not per pair
                v = grid_atlas.get_submap(a).features[0].vertex_positions()
                Ra, ta = grid_atlas.updated_submap_pose(a)
                world_pts[a] = v.to(Ra.device) @ Ra.T + ta.reshape(1, 3)
            if b not in inv:
                Rb, tb = grid_atlas.updated_submap_pose(b)
                inv[b] = (Rb, tb.reshape(1, 3), grid_atlas.get_submap(b).bound.to(Rb))
            Rb, tb, bd = inv[b]
            q = (world_pts[a] - tb) @ Rb
            fracs.append(((q >= bd[:, 0]) & (q <= bd[:, 1])).all(dim=1).float().mean())
        fracs = torch.stack(fracs).cpu().tolist()
    # quantised: every rank must arrive at the SAME deal (partition_pairs sorts by cost), and two GPUs may differ in the
    # last bits of a mean -- a pair dropped or counted twice would go unnoticed in the all-reduced gradient
    return [1.0 + 30.0 * (round(f * 1024.0) / 1024.0) for f in fracs]


# Calibration of the iteration-time estimate (1xMI355X, cfg-4: 8 ScanNet-shaped submaps, 28 pairs, profiles/r05_bench.json):
# level 0 (0.9 M source vertices) 55 us, level 1 (112 M source vertices, 21 % in bound) 578 us per iteration.  In the cost
# unit of pair_costs (1 per source vertex, 30 more per in-bound one) level 1 is 112e6 x (1 + 30 x 0.21) = 8.2e8 units for
# ~530 us of pair stage: 0.65e-6 us per unit, on top of ~50 us that do not depend on the pair list (three launches, the
# gate, the epilogues).  Only the first part shrinks when the pairs are dealt over ranks.  (Round 4: 1e-6 and 75.)
# Round 6 (profiles/r06_bench.json): level 1 464 us with the brick-ordered vertices -> ~415 us of pair stage: 0.51e-6.
PAIR_US_PER_UNIT = 0.51e-6
ITERATION_FIXED_US = 50.0
SHARD_OVERHEAD_US = 15.0       # two graph replays per iteration instead of one eighth of an 8x unrolled one, + the hook


def pair_stage_estimate_us(grid_atlas, pairs, costs, level) -> float:
    """Estimated time of ONE rank's pair stage over `pairs` (us): source vertices of the level x pair_costs."""
    units = 0.0
    for (a, _), c in zip(pairs, costs):
        units += float(grid_atlas.coordinates_for_alignment(a, level).shape[0]) * c
    return PAIR_US_PER_UNIT * units


_ALLREDUCE_US = {}


def measured_all_reduce_us(n_floats: int, device) -> float:
    """Latency of one all-reduce(SUM) of n_floats fp32 on this process group (us), probed once per (backend, world,
    device type): 5 warm-up + 20 timed collectives on a scratch buffer, then MAX over ranks so that every rank holds the
    same figure (the policy below must come out the same everywhere).  A collective: every rank must call it."""
    import time
    key = (dist.get_backend(), dist.get_world_size(), torch.device(device).type, int(n_floats))
    if key in _ALLREDUCE_US:
        return _ALLREDUCE_US[key]
    buf = torch.zeros(int(n_floats), device=device)
    sync = torch.cuda.synchronize if buf.is_cuda else (lambda: None)
    for _ in range(5):
        all_reduce_sum(buf)
    sync()
    t0 = time.perf_counter()
    for _ in range(20):
        all_reduce_sum(buf)
    sync()
    us = torch.tensor([(time.perf_counter() - t0) / 20 * 1e6], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        us = us.to(device)
    dist.all_reduce(us, op=dist.ReduceOp.MAX)
    _ALLREDUCE_US[key] = float(us.item())
    return _ALLREDUCE_US[key]


def alignment_mode(pair_stage_us: float, all_reduce_us: float, world: int, mode: Optional[str] = None) -> str:
    """'replicated' or 'sharded' for one alignment level.  Sharding divides the pair stage by `world` (at best) and adds
    an all-reduce and a second graph replay to EVERY iteration; it is taken only when that is a gain:

        pair_stage_us (1 - 1 / world)  >  all_reduce_us + SHARD_OVERHEAD_US

    cfg-4 on 8 ranks with a ~25 us RCCL all-reduce: level 0 (7 us of pair stage) stays replicated -- sharded it would run
    at ~1.5x the single-GPU time -- and level 1 (800 us) is sharded.  mode / MISO_ALIGN_DIST_MODE force either."""
    mode = mode or os.environ.get("MISO_ALIGN_DIST_MODE") or "auto"
    if mode in ("replicated", "sharded"):
        return mode
    if mode != "auto":
        raise ValueError(f"alignment mode {mode!r}: expected auto, replicated or sharded")
    if world <= 1:
        return "replicated"
    return "sharded" if pair_stage_us * (1.0 - 1.0 / world) > all_reduce_us + SHARD_OVERHEAD_US else "replicated"


def _deal(pairs, costs, world) -> List[int]:
    """owner[i] of every pair: longest first onto the least loaded rank (ties: lower rank)."""
    load = [0.0] * world
    owner = [0] * len(pairs)
    for i in sorted(range(len(pairs)), key=lambda i: (-costs[i], i)):
        k = min(range(world), key=lambda k: (load[k], k))
        owner[i] = k
        load[k] += costs[i]
    return owner


def agree_on_plan(owner: Sequence[int], mode: str, costs: Sequence[float], device) -> Tuple[List[int], str, bool]:
    """Rank 0's deal and mode become everybody's: ONE broadcast of len(owner) + 1 integers.  Every rank derives costs
    from its own GPU's floats, and although they are quantised a mean that sits at a rounding boundary can still come out
    one step apart on two devices -- a pair evaluated twice or by nobody would vanish in the all-reduced gradient
    without a trace.  So the ranks do not each trust their own deal.  Returns (owner, mode, costs_agree): the last says
    whether every rank had computed the same quantised costs (an all-reduce MIN / MAX of a checksum) -- False is logged,
    not fatal: the broadcast plan is consistent either way."""
    rank, world = rank_world()
    on_dev = dist.get_backend() == "nccl"
    t = torch.tensor(list(owner) + [1 if mode == "sharded" else 0], dtype=torch.int64)
    t = t.to(device) if on_dev else t
    dist.broadcast(t, src=0)
    vals = t.cpu().tolist()
    # checksum of the quantised costs (exact integers: costs are 1 + 30 k / 1024)
    q = [int(round((c - 1.0) / 30.0 * 1024.0)) for c in costs]
    h = 0
    for i, v in enumerate(q):
        h = (h * 1000003 + v * 31 + i) % ((1 << 53) - 111)
    cs = torch.tensor([float(h), -float(h)], dtype=torch.float64)
    cs = cs.to(device) if on_dev else cs
    dist.all_reduce(cs, op=dist.ReduceOp.MAX)
    agree = float(cs[0].item()) == -float(cs[1].item())
    if not agree and rank == 0:
        logger.warning("pair costs differ between ranks (device-dependent rounding of the overlap estimate); "
                       "rank 0's deal is used everywhere")
    return [int(v) for v in vals[:-1]], ("sharded" if vals[-1] else "replicated"), agree


def partition_pairs(pairs: Sequence[Tuple[int, int]], rank: Optional[int] = None,
                    world: Optional[int] = None, costs: Optional[Sequence[float]] = None) -> List[Tuple[int, int]]:
    """This rank's share of the pair list.  Without costs: round-robin.  With costs (pair_costs): longest first onto
    the least loaded rank (ties: lower rank), the share returned in list order -- every rank computes the same deal."""
    r, w = rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    if costs is None:
        return [p for i, p in enumerate(pairs) if i % world == rank]
    assert len(costs) == len(pairs)
    owner = _deal(pairs, costs, world)
    return [p for i, p in enumerate(pairs) if owner[i] == rank]


def align_multiple_submaps_distributed(grid_atlas, dataset, pairwise_loss_tuple, num_iters=10, lr=1e-2,
                                       rel_change_thresh=0, submap_pairs=None, check_intersection=True,
                                       pose_reg_weight=0, pose_thresh_rad=1.0, pose_thresh_m=1.0,
                                       verbose=False, save_iterations=False, always_reduce=False, mode=None):
    """generic_align_multiple_submaps (grid_opt/align/base.py:89-163) with the pair list sharded over ranks.
    Every rank must hold all submaps (sync_submaps) and identical pose parameters.

    The loop is base.fused_alignment_loop: this rank's pairs go through ONE overlap launch and ONE pair launch per
    iteration (no per-pair Python, no host sync on the overlap test or the NaN guard), the 7S + 1 floats of pose
    gradients + loss are all-reduced, and the identical regulariser / NaN guard / Adam step runs on every rank, so
    the replicas stay bit-identical.  Results equal the single-process run up to fp32 summation order of the pair
    sums.  A pair loss that carries ``fused`` (align.miso.latent_loss_for_level) runs as that loop; one that does not
    (the SDF fine-tune stage) runs the op-by-op loop with the same sharding (generic_align_multiple_submaps, my_pairs /
    reduce).  always_reduce: keep the all-reduce hook live in a group of one rank (exercises RCCL on a single GPU).

    mode ('auto' | 'replicated' | 'sharded'; default auto, or MISO_ALIGN_DIST_MODE): a fused level whose pair stage is
    cheaper than the collective it would need runs REPLICATED -- every rank evaluates all pairs with the single-process
    loop (one graph, 8x unrolled), no collective inside the loop, bit-identical replicas by construction -- so that adding
    GPUs can never make a level slower than one GPU (``alignment_mode``).  The decision and the deal are rank 0's,
    broadcast once per call (``agree_on_plan``).  The returned dict carries ``dist``: mode, the estimate, the measured
    all-reduce latency, this rank's pairs, and whether the ranks' own cost estimates agreed."""
    from miso_amd.grid_opt.align.base import fused_alignment_loop
    rank, world = rank_world()
    loss_name, loss_func = pairwise_loss_tuple
    fused = getattr(loss_func, 'fused', None)
    if submap_pairs is None:
        n = grid_atlas.num_submaps
        submap_pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    if fused is None:
        # a pair loss without a fused plan -- the SDF fine-tune stage (align/miso.py pairwise_loss_sdf; reference
        # miso.py:283-319, `--use_sdf`): the op-by-op loop with the pair list dealt round-robin, every rank evaluating its
        # pairs through autograd, one all-reduce of the pose gradients + loss per iteration, the same Adam step everywhere
        from miso_amd.grid_opt.align.base import generic_align_multiple_submaps
        red = all_reduce_sum if world > 1 else ((lambda t: all_reduce_sum(t, always=True)) if always_reduce and
                                                 dist.is_available() and dist.is_initialized() else None)
        return generic_align_multiple_submaps(grid_atlas, dataset, pairwise_loss_tuple, num_iters=num_iters, lr=lr,
                                              rel_change_thresh=rel_change_thresh, submap_pairs=submap_pairs,
                                              check_intersection=check_intersection, pose_reg_weight=pose_reg_weight,
                                              pose_thresh_rad=pose_thresh_rad, pose_thresh_m=pose_thresh_m,
                                              verbose=verbose and rank == 0, save_iterations=save_iterations,
                                              my_pairs=partition_pairs(submap_pairs, rank, world), reduce=red)
    level = fused.get('level', 0)
    info = {'mode': 'replicated', 'world': world, 'pairs_this_rank': len(submap_pairs)}
    my_pairs = list(submap_pairs)
    if world > 1:
        dev = grid_atlas.rotation_corrections[0].device
        costs = pair_costs(grid_atlas, submap_pairs)
        est_us = pair_stage_estimate_us(grid_atlas, submap_pairs, costs, level)
        ar_us = measured_all_reduce_us(7 * grid_atlas.num_submaps + 1, dev)
        owner, want, agree = agree_on_plan(_deal(submap_pairs, costs, world), alignment_mode(est_us, ar_us, world, mode),
                                           costs, dev)
        info.update(mode=want, pair_stage_estimate_us=est_us, all_reduce_us=ar_us, costs_agree=agree)
        if want == 'sharded':
            my_pairs = [p for i, p in enumerate(submap_pairs) if owner[i] == rank]
            info['pairs_this_rank'] = len(my_pairs)
    timer = utils.PerfTimer(activate=True)
    reduce = None
    if world > 1 and info['mode'] == 'sharded':
        reduce = all_reduce_sum
    elif world == 1 and always_reduce and dist.is_available() and dist.is_initialized():      # the single-GPU RCCL smoke test
        reduce = lambda t: all_reduce_sum(t, always=True)                       # noqa: E731
    iteration_results = fused_alignment_loop(grid_atlas, fused, submap_pairs, check_intersection, lr, num_iters,
                                             rel_change_thresh, pose_reg_weight, pose_thresh_rad, pose_thresh_m,
                                             verbose and rank == 0, save_iterations, f"{loss_name}[rank {rank}/{world}]",
                                             reduce=reduce, my_pairs=my_pairs if reduce is not None else None)
    if world > 1 and info['mode'] == 'replicated':
        # Replicated ranks run the same loop on the same inputs, but the pair sums are fp64 atomics whose order differs from
        # run to run (reproducible to ~1e-16, i.e. the fp32 gradients to an occasional ulp): after Adam the replicas can
        # drift by an ulp, or stop one iteration apart under rel_change_thresh, and nothing inside a replicated loop brings
        # them back (the sharded loop's all-reduce does).  One tiny collective per CALL, outside the loop: rank 0's final
        # corrections go to everybody.
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in list(grid_atlas.rotation_corrections) +
                              list(grid_atlas.translation_corrections)])
            _broadcast(flat, 0)
            off = 0
            for p in list(grid_atlas.rotation_corrections) + list(grid_atlas.translation_corrections):
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
    cpu_time, gpu_time = timer.check()
    return {'cpu_time_sec': cpu_time, 'gpu_time_sec': gpu_time, 'iteration_results': iteration_results, 'dist': info}
