"""world_size-2 tests of the submap-parallel layer on CPU (gloo).  The HIP operators are
replaced by the oracle stand-ins (tests/oracle_backend.py) inside every worker: what is
tested here is the sharding / collective logic, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import oracle_backend
    oracle_backend.install(_Patch())
    import golden_cases as gc
    from test_grid_opt_mirror import make_atlas
    from miso_amd import dist as mdist
    import miso_amd.grid_opt.align.miso as AM
    r, w = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    c = gc.ATLAS
    atlas = make_atlas("cpu")
    # --- mapping shard: every rank perturbs ONLY its own submaps, then sync ----------------
    assert mdist.owned_submaps(3) == ([0, 2] if rank == 0 else [1])

    def map_one(s):
        with torch.no_grad():
            for g in atlas.get_submap(s).features:
                g.feature.add_(0.01 * (s + 1))
    before = [[g.feature.detach().clone() for g in atlas.get_submap(s).features] for s in range(3)]
    mdist.map_submaps_parallel(atlas, map_one, sync=True)
    for s in range(3):
        for g, b in zip(atlas.get_submap(s).features, before[s]):
            torch.testing.assert_close(g.feature.detach(), b + 0.01 * (s + 1))
            assert g.feature.is_contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad():   # undo, so that the alignment below matches the golden atlas
        for s in range(3):
            for g, b in zip(atlas.get_submap(s).features, before[s]):
                g.feature.copy_(b)
    # --- alignment: pairs sharded, one all-reduce per iteration ------------------------------
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    assert mdist.partition_pairs([(0, 1), (0, 2), (1, 2)]) == ([(0, 1), (1, 2)] if rank == 0 else [(0, 2)])

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return 0

    res = {}
    start = [(atlas.rotation_corrections[s].detach().clone(), atlas.translation_corrections[s].detach().clone())
             for s in range(3)]
    for l in range(c["n_levels"]):
        # the sharded loop is the fused one (ops.AlignPlan; here its oracle stand-in): this rank's pairs in one
        # launch, ONE all-reduce of 6S + 1 floats per iteration, identical guard / Adam on every rank
        tup = (f"latent{l}", AM.latent_loss_for_level(atlas, l, align_loss="L2", device="cpu"))
        info = mdist.align_multiple_submaps_distributed(atlas, DS(), tup, num_iters=3, lr=1e-2, pose_reg_weight=1.0,
                                                        pose_thresh_rad=1e-3, pose_thresh_m=1e-3,
                                                        save_iterations=(l == 0), mode="sharded")
        assert info["dist"]["mode"] == "sharded" and info["dist"]["costs_agree"]
        assert info["dist"]["pairs_this_rank"] in (1, 2) and info["dist"]["all_reduce_us"] > 0
        if l == 0:
            assert sorted(info["iteration_results"]) == [0, 1, 2, 3]
            res["snap0"] = torch.stack([info["iteration_results"][i] for i in range(4)]).numpy()
        res[f"dr{l}"] = torch.stack([p.detach() for p in atlas.rotation_corrections]).numpy()
        res[f"dt{l}"] = torch.stack([p.detach() for p in atlas.translation_corrections]).numpy()
    # --- the policy: a level this small must come out REPLICATED (every rank all pairs, no collective in the loop) ------
    after = [(atlas.rotation_corrections[s].detach().clone(), atlas.translation_corrections[s].detach().clone())
             for s in range(3)]

    def restart():
        for s in range(3):
            atlas.set_submap_pose_correction(s, *start[s])
    restart()
    calls = []
    real_reduce = mdist.all_reduce_sum
    mdist.all_reduce_sum = lambda t, always=False: (calls.append(t.numel()), real_reduce(t, always))[1]
    try:
        for l in range(c["n_levels"]):
            tup = (f"latent{l}", AM.latent_loss_for_level(atlas, l, align_loss="L2", device="cpu"))
            info = mdist.align_multiple_submaps_distributed(atlas, DS(), tup, num_iters=3, lr=1e-2, pose_reg_weight=1.0,
                                                            pose_thresh_rad=1e-3, pose_thresh_m=1e-3)
            assert info["dist"]["mode"] == "replicated", info["dist"]      # a few thousand vertices against a collective
            assert info["dist"]["pairs_this_rank"] == 3
            res[f"rep_dr{l}"] = torch.stack([p.detach() for p in atlas.rotation_corrections]).numpy()
            res[f"rep_dt{l}"] = torch.stack([p.detach() for p in atlas.translation_corrections]).numpy()
        assert not [n for n in calls if n == 7 * 3 + 1], "a replicated level must not all-reduce pose gradients"
    finally:
        mdist.all_reduce_sum = real_reduce
    # --- ranks that disagree on the costs: detected, and harmless (rank 0's deal is everybody's) --------------------
    restart()
    real_costs = mdist.pair_costs
    if rank == 1:       # what a rounding boundary on another GPU would do: one pair a quantum cheaper, the order of the deal flips
        mdist.pair_costs = lambda at, pairs: [c_ - (30.0 / 1024.0 if i == 0 else 0.0) + (30.0 * (i == 2))
                                              for i, c_ in enumerate(real_costs(at, pairs))]
    try:
        for l in range(c["n_levels"]):
            tup = (f"latent{l}", AM.latent_loss_for_level(atlas, l, align_loss="L2", device="cpu"))
            info = mdist.align_multiple_submaps_distributed(atlas, DS(), tup, num_iters=3, lr=1e-2, pose_reg_weight=1.0,
                                                            pose_thresh_rad=1e-3, pose_thresh_m=1e-3, mode="sharded")
            assert info["dist"]["costs_agree"] is False
            res[f"dis_dr{l}"] = torch.stack([p.detach() for p in atlas.rotation_corrections]).numpy()
            res[f"dis_dt{l}"] = torch.stack([p.detach() for p in atlas.translation_corrections]).numpy()
            res[f"dis_pairs{l}"] = np.array([info["dist"]["pairs_this_rank"]])
    finally:
        mdist.pair_costs = real_costs
    for s in range(3):
        atlas.set_submap_pose_correction(s, *after[s])
    # --- SDF fine-tune stage (reference align/miso.py:283-319, --use_sdf): a pair loss WITHOUT a fused plan, sharded ---
    res.update(_sdf_stage(mdist.align_multiple_submaps_distributed))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    torch.distributed.destroy_process_group()


def _sdf_stage(align):
    """Three iterations of the SDF-space alignment of the two-keyframe atlas on its fixed batch; `align` is the sharded
    entry point (workers) or the single-process loop (reference run)."""
    import golden_cases as gc
    from test_grid_opt_mirror import make_atlas_two_kf, _OneBatch
    import miso_amd.grid_opt.align.miso as AM
    atlas = make_atlas_two_kf("cpu")
    mi, gt = gc.atlas_sdf_batch()
    ds = _OneBatch(mi, gt)

    def sdf(at, loader, a, b):
        return AM.pairwise_loss_sdf(at, loader, a, b, align_loss="L2", device="cpu")
    info = align(atlas, ds, ("hier_sdf_L2", sdf), num_iters=2, lr=1e-3, pose_reg_weight=1.0, pose_thresh_rad=1e-3,
                 pose_thresh_m=1e-3, verbose=False, check_intersection=False)
    assert set(info) >= {"cpu_time_sec", "gpu_time_sec", "iteration_results"}
    return {"sdf_dr": torch.stack([p.detach() for p in atlas.rotation_corrections]).numpy(),
            "sdf_dt": torch.stack([p.detach() for p in atlas.translation_corrections]).numpy()}


@pytest.mark.timeout(300)
def test_submap_parallel_world2_matches_single_process(tmp_path, monkeypatch):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz")
    b = np.load(tmp_path / "rank1.npz")
    for k in a.files:
        if k.startswith("dis_pairs"):
            assert int(a[k][0]) + int(b[k][0]) == 3   # ranks that disagreed on the costs still deal every pair exactly once
            continue
        np.testing.assert_array_equal(a[k], b[k])     # replicas stay bit-identical
    # single-process reference run of the same loop (with the same regulariser)
    sys.path.insert(0, HERE)
    import oracle_backend
    oracle_backend.install(monkeypatch)
    import golden_cases as gc
    from test_grid_opt_mirror import make_atlas
    import miso_amd.grid_opt.align.base as AB
    import miso_amd.grid_opt.align.miso as AM
    c = gc.ATLAS
    atlas = make_atlas("cpu")
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return 0

    for l in range(c["n_levels"]):
        # op-by-op loop: autograd through so3_exp_map, one loss per pair, DenseAdam
        tup = (f"latent{l}", lambda at, ld, x, y, _l=l: AM.pairwise_loss_latent(
            at, ld, x, y, level=_l, fdim=c["fdim"], align_loss="L2", device="cpu"))
        info = AB.generic_align_multiple_submaps(atlas, DS(), tup, num_iters=3, lr=1e-2, verbose=False,
                                                 pose_reg_weight=1.0, pose_thresh_rad=1e-3, pose_thresh_m=1e-3,
                                                 save_iterations=(l == 0))
        if l == 0:
            snap = torch.stack([info["iteration_results"][i] for i in range(4)]).numpy()
            np.testing.assert_allclose(a["snap0"], snap, rtol=0, atol=2e-5)
        dr = torch.stack([p.detach() for p in atlas.rotation_corrections]).numpy()
        dt = torch.stack([p.detach() for p in atlas.translation_corrections]).numpy()
        np.testing.assert_allclose(a[f"dr{l}"], dr, rtol=0, atol=2e-5)
        np.testing.assert_allclose(a[f"dt{l}"], dt, rtol=0, atol=2e-5)
        # world-2 replicated == world-2 sharded == single process; a deal made under disagreeing costs too
        for tag in ("rep", "dis"):
            np.testing.assert_allclose(a[f"{tag}_dr{l}"], dr, rtol=0, atol=2e-5)
            np.testing.assert_allclose(a[f"{tag}_dt{l}"], dt, rtol=0, atol=2e-5)
    # the SDF fine-tune stage: the sharded op-by-op loop == the single-process one (poses moved, and by the same amounts)
    ref = _sdf_stage(AB.generic_align_multiple_submaps)
    assert np.abs(ref["sdf_dr"]).max() > 1e-4
    np.testing.assert_allclose(a["sdf_dr"], ref["sdf_dr"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(a["sdf_dt"], ref["sdf_dt"], rtol=0, atol=2e-6)


def test_alignment_mode_policy():
    """Sharding must buy more than it costs: the pair stage divided by the ranks against one all-reduce + a second
    replay per iteration.  cfg-4's level 0 (a few us of pair stage) stays replicated on any world size; level 1 is sharded."""
    from miso_amd import dist as mdist
    assert mdist.alignment_mode(7.0, 25.0, 8, "auto") == "replicated"
    assert mdist.alignment_mode(800.0, 25.0, 8, "auto") == "sharded"
    assert mdist.alignment_mode(800.0, 25.0, 1, "auto") == "replicated"
    assert mdist.alignment_mode(60.0, 25.0, 2, "auto") == "replicated"      # 30 us saved < 40 us added
    assert mdist.alignment_mode(7.0, 25.0, 8, "sharded") == "sharded" and mdist.alignment_mode(800.0, 25.0, 8, "replicated") == "replicated"
    with pytest.raises(ValueError):
        mdist.alignment_mode(1.0, 1.0, 2, "sometimes")


def test_shard_helpers():
    from miso_amd import dist as mdist
    assert mdist.owned_submaps(8, rank=3, world=8) == [3]
    assert mdist.owned_submaps(8, rank=1, world=4) == [1, 5]
    assert mdist.owned_submaps(3, rank=0, world=1) == [0, 1, 2]
    pairs = [(a, b) for a in range(8) for b in range(a + 1, 8)]
    parts = [mdist.partition_pairs(pairs, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == pairs and max(map(len, parts)) - min(map(len, parts)) <= 1
    # dealt by cost: a gated pair (cost 1) is nearly free next to an overlapping one (cost ~ 1 + 30 x in-bound
    # fraction); longest first onto the least loaded rank -- every pair exactly once, loads within one heavy pair
    costs = [1.0 + 30.0 * (0.5 if (b - a) in (1, 2) else 0.0) for a, b in pairs]
    parts = [mdist.partition_pairs(pairs, r, 4, costs=costs) for r in range(4)]
    assert sorted(sum(parts, [])) == pairs
    load = [sum(costs[pairs.index(p)] for p in part) for part in parts]
    assert max(load) - min(load) <= 16.0 + 1e-9
    rr = [sum(costs[i] for i in range(len(pairs)) if i % 4 == r) for r in range(4)]
    assert max(load) <= max(rr)                      # never worse than the round-robin deal
    assert all(part == sorted(part, key=pairs.index) for part in parts)      # a share keeps the list order
