"""Submap-parallel execution over ``torch.distributed`` -- one process per GPU, backend
``nccl`` (= RCCL over xGMI on MI355X), ``gloo`` in CPU tests.

The reference is single-process (no distributed code at all); the shard
unit follows from its data model: every submap is an independent GridNet with its own
grids, keyframe poses and optimiser state (grid_opt/models/grid_atlas.py:146-150).

* mapping   rank r owns submaps {s : s % world == r}; NO per-step collective (decoder
            frozen, grids disjoint).  ``sync_submaps`` broadcasts each owner's grids
            once afterwards (61 MiB per ScanNet submap ~ 0.4 ms per xGMI hop).
* alignment grids are read-only and replicated; the pair list is dealt round-robin;
            each iteration ends with ONE all-reduce(SUM) of a flat fp32 buffer holding
            the pose gradients of submaps 1..S-1 and the loss (6(S-1)+1 floats: latency
            bound, xGMI bandwidth irrelevant), then the identical Adam step everywhere.
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


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks of a small flat tensor (the 6S + 1 floats of an alignment iteration)."""
    if rank_world()[1] == 1:
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


@torch.no_grad()
def sync_submaps(atlas):
    """Every rank ends up with every submap's parameters and buffers (owner broadcasts)."""
    _, world = rank_world()
    if world == 1:
        return
    for s in range(atlas.num_submaps):
        src = owner_of(s, world)
        sub = atlas.get_submap(s)
        for t in list(sub.parameters()) + list(sub.buffers()):
            flat = _flat_view(t.data)
            _broadcast(flat, src)
            if flat.data_ptr() != t.data.data_ptr():   # a copy was needed: write it back
                t.data.copy_(flat.view_as(t.data))


def partition_pairs(pairs: Sequence[Tuple[int, int]], rank: Optional[int] = None,
                    world: Optional[int] = None) -> List[Tuple[int, int]]:
    r, w = rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    return [p for i, p in enumerate(pairs) if i % world == rank]


def align_multiple_submaps_distributed(grid_atlas, dataset, pairwise_loss_tuple, num_iters=10, lr=1e-2,
                                       rel_change_thresh=0, submap_pairs=None, check_intersection=True,
                                       pose_reg_weight=0, pose_thresh_rad=1.0, pose_thresh_m=1.0,
                                       verbose=False, save_iterations=False):
    """generic_align_multiple_submaps (grid_opt/align/base.py:89-163) with the pair list sharded over ranks.
    Every rank must hold all submaps (sync_submaps) and identical pose parameters.

    The loop is base.fused_alignment_loop: this rank's pairs go through ONE overlap launch and ONE pair launch per
    iteration (no per-pair Python, no host sync on the overlap test or the NaN guard), the 6S + 1 floats of pose
    gradients + loss are all-reduced, and the identical regulariser / NaN guard / Adam step runs on every rank, so
    the replicas stay bit-identical.  Results equal the single-process run up to fp32 summation order of the pair
    sums.  The pair loss must carry ``fused`` (align.miso.latent_loss_for_level)."""
    from miso_amd.grid_opt.align.base import fused_alignment_loop
    rank, world = rank_world()
    loss_name, loss_func = pairwise_loss_tuple
    fused = getattr(loss_func, 'fused', None)
    if fused is None:
        raise ValueError("align_multiple_submaps_distributed needs a fused pair loss "
                         "(miso_amd.grid_opt.align.miso.latent_loss_for_level); the SDF fine-tune stage is not sharded")
    if submap_pairs is None:
        n = grid_atlas.num_submaps
        submap_pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    my_pairs = partition_pairs(submap_pairs, rank, world)
    timer = utils.PerfTimer(activate=True)
    reduce = all_reduce_sum if world > 1 else None
    iteration_results = fused_alignment_loop(grid_atlas, fused, submap_pairs, check_intersection, lr, num_iters,
                                             rel_change_thresh, pose_reg_weight, pose_thresh_rad, pose_thresh_m,
                                             verbose and rank == 0, save_iterations, f"{loss_name}[rank {rank}/{world}]",
                                             reduce=reduce, my_pairs=my_pairs)
    cpu_time, gpu_time = timer.check()
    return {'cpu_time_sec': cpu_time, 'gpu_time_sec': gpu_time, 'iteration_results': iteration_results}
