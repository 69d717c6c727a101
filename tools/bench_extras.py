"""The secondary figures of bench.py (SURVEY 8d beside the headline; VERDICT r5 hygiene: split out of bench.py, which
re-exports them): cfg-3 / cfg-4 / cfg-5 workloads, trainer steps with rooflines, sample generation, mesh extraction
(GridNet and atlas), the eikonal step, SLAM components, and the multi-rank extras.  The headline, its roofline, the CPU
baseline and the launch protocol stay in bench.py."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _b():
    """bench.py's constants and helpers (imported late: bench.py imports this module at its end)."""
    import bench
    return bench


class _Lazy:
    """N_POINTS, LEVELS, C, HIDDEN, time_kernel, ... resolved in bench.py on first use"""

    def __getattr__(self, k):
        return getattr(_b(), k)


B = _Lazy()

SCANNET_CFG = {"name": "grid_net", "spatial_dim": 3,
               "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                           "fix": True, "pretrained_model": None},
               "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 1e-2,
                        "bound": [[-10., 10.], [-5., 5.], [-10., 10.]], "base_cell_size": 0.5, "per_level_scale": 5,
                        "n_levels": 2},
               "pose": {"optimize": False, "num_poses": 1}}


def scannet_atlas(dev, n_submaps=8, perturb=True):
    """BASELINE configs 3 / 4: n_submaps ScanNet-shaped submaps (bound 20x10x20 m, cells 0.5 / 0.1 m,
    C=4: 40x20x40 + 200x100x200 per submap), identity rotations, translations on a 2 x (n/2) lattice with 50 %
    overlap; pose corrections perturbed by up to 10 deg / 0.5 m (seed 55, as demo/align_submaps.py:241-242,267-273).
    Every rank builds the same atlas from the same seeds."""
    import math
    from miso_amd.grid_opt.models.grid_atlas import GridAtlas
    torch.manual_seed(0)
    atlas = GridAtlas(SCANNET_CFG, device=dev)
    lb = torch.tensor(SCANNET_CFG["grid"]["bound"])
    for s in range(n_submaps):
        tx, tz = 10.0 * (s // 2), 10.0 * (s % 2)
        atlas.add_submap(lb, torch.eye(3), torch.tensor([[tx], [0.0], [tz]]), num_poses=1)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
    atlas.to(dev)
    if perturb:
        g = torch.Generator().manual_seed(55)
        for s in range(1, n_submaps):
            axis = torch.randn(3, generator=g)
            axis = axis / axis.norm()
            dr = axis * math.radians(10.0) * torch.rand(1, generator=g)
            dt = (torch.rand(3, generator=g) * 2 - 1) * 0.5
            atlas.set_submap_pose_correction(s, dr.reshape(1, 3).to(dev), dt.reshape(3, 1).to(dev))
    return atlas


def cfg4_pmc_traffic(level):
    """HBM bytes per launch of pair_stage_kernel (gate + residual in one launch; pair_latent_batch_kernel where the stage
    is split) at an alignment level, from the committed PMC summary -- only
    while that summary was collected on the kernel sources of this library (the hash miso_version() embeds)."""
    import glob
    from miso_amd.csrc_hash import source_hash
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            js = json.load(open(pmc))
        except Exception:
            continue
        if js.get("_meta", {}).get("source_hash") == source_hash():
            e = js.get("cfg4_align", {}).get(f"pair_stage_kernel_level{level}") or \
                js.get("cfg4_align", {}).get(f"pair_latent_batch_kernel_level{level}")
            # ONE number: tools/ubench/fetch_calib.hip read known byte counts under the counter -- a
            # coalesced stream, 16-B rows gathered at random and at a 256-B stride, all from HBM: every fill is a 128-B
            # request (TCC_EA0_RDREQ_128B) and FETCH_SIZE reports 64 B of it, for the gathers exactly as for the stream
            return None if e is None else {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
                                           "calibration": "2 x FETCH_SIZE KiB + WRITE_SIZE KiB; the x2 holds for 16-byte gathers "
                                                          "as for streams (tools/ubench/fetch_calib.hip, tools/fetch_calib.sh)",
                                           "source": os.path.basename(pmc)}
    return None


def align_cfg4(dev, atlas, levels=(0, 1), iters=20, dist=None):
    """cfg-4: latent alignment of all S(S-1)/2 pairs, generic_align_multiple_submaps with the reference's own
    alignment settings (verbose + save_iterations, configs/rgbd/scannet.yaml:65-66).  Wall time per iteration of the
    fused loop; with a process group the pair list is sharded (miso_amd.dist) and each iteration has ONE all-reduce
    of 6S + 1 floats."""
    import miso_amd.grid_opt.align.base as AB
    import miso_amd.grid_opt.align.miso as AM
    from miso_amd import dist as mdist
    S = atlas.num_submaps
    out = {"submaps": S, "pairs": S * (S - 1) // 2, "iterations_timed": iters}
    start = [(atlas.rotation_corrections[s].detach().clone(), atlas.translation_corrections[s].detach().clone())
             for s in range(S)]

    class _DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return 0

    for level in levels:
        loss = AM.latent_loss_for_level(atlas, level, device=dev)

        dist_info = {}

        def run(n_it, lr=1e-2, one_rank=False):
            for s in range(S):
                atlas.set_submap_pose_correction(s, *start[s])
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            if dist is not None and not one_rank:
                info = mdist.align_multiple_submaps_distributed(atlas, _DS(), (f"latent{level}", loss),
                                                                num_iters=n_it - 1, lr=lr, verbose=True,
                                                                save_iterations=True)
                dist_info.update(info["dist"])
            else:
                AB.generic_align_multiple_submaps(atlas, _DS(), (f"latent{level}", loss), num_iters=n_it - 1, lr=lr,
                                                  verbose=True, save_iterations=True)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            lp = atlas.__dict__["_last_align_loop"]      # the iterations proper (set-up, warm-up, capture excluded)
            return time.perf_counter() - t0, lp["seconds"] / max(lp["iterations"], 1)

        run(4)                                      # warm-up (plan build, source features cached)
        t_a = min(run(iters)[0] for _ in range(3))
        per_it = min(run(6 * iters)[1] for _ in range(2))
        # the same with lr = 0: Adam moves nothing, every iteration sees the START poses -- the state the pair stage
        # below is timed at (while the poses move the overlap changes, and so does the work per iteration)
        per_it_fixed = min(run(6 * iters, 0.0)[1] for _ in range(2))
        nv = sum(atlas.coordinates_for_alignment(a, level).shape[0] for a in range(S) for b in range(a + 1, S))
        C_ = SCANNET_CFG["grid"]["feature_dim"]
        rec = {"source_vertices_per_iteration": nv, "ms_per_iteration": per_it * 1e3,
               "ms_per_iteration_fixed_start_poses": per_it_fixed * 1e3,
               f"ms_{iters}_iterations": t_a * 1e3, "vertices_per_s": nv / per_it,
               "ms_per_pair_iteration": per_it * 1e3 / out["pairs"]}
        if dist is None:
            # pair kernel alone, all pairs in one launch (HIP events), against its algorithmic bytes: 12 B of
            # coordinates per source vertex + for the in-bound ones 4 C (level+1) B of source features and
            # 32 C (level+1) B of gathered destination corners
            from miso_amd import ops
            pairs = [(a, b) for a in range(S) for b in range(a + 1, S)]
            R0 = torch.stack(list(atlas.R_world_submap_list))
            t0_ = torch.stack(list(atlas.t_world_submap_list))
            plan = ops.AlignPlan(R0, t0_, loss.fused["inputs"](atlas, pairs, True), ring_iters=1)
            plan.params.copy_(torch.cat((torch.cat([r.reshape(1, 3) for r, _ in start]),
                                         torch.cat([t.reshape(1, 3) for _, t in start])), 1))
            t_k = B.time_kernel(plan.iteration_a, iters=10, warm=2)
            inb = float(plan.pair_out[:, 1].sum().item())
            b_alg = 12 * nv + inb * (4 + 32) * C_ * (level + 1)
            rec["pair_stage_us"] = t_k
            rec["pair_stage_note"] = ("iteration_a (prologue, gates, pair kernel, epilogue A) at the start poses, launch by "
                                      "launch with HIP events: compare with ms_per_iteration_fixed_start_poses (the "
                                      "captured loop at the same poses)")
            rec["in_bound_vertices"] = inb
            # COMPULSORY bytes: what has to come from HBM at least once per pair -- the source's vertices (12 B) and its
            # feature rows for the in-bound ones, and the part of the destination's levels the source overlaps (an
            # in-bound vertex touches 8 corners, but a lattice shares them: one row of 4 C B per destination vertex in
            # the overlap, ~ one per in-bound source vertex per level at equal cell sizes).  VERDICT r3 item 3: `frac` is
            # on these; the no-reuse figure (8 corner fetches per in-bound vertex, SURVEY 8d's convention) is kept beside it.
            # Round 5: the kernel no longer reads every source vertex -- a box per 64 vertices lets it skip the runs that
            # cannot reach the destination bound (miso_align_src_boxes) -- so the coordinates that HAVE to be read are those
            # of the in-bound vertices (12 B each) plus the box table (24 B per 64 vertices); what the kernel still reads
            # beyond that (vertices of runs its conservative test cannot rule out) is its own choice and not counted.
            n_runs = sum((atlas.coordinates_for_alignment(a, level).shape[0] + 63) // 64 for a, b in pairs)
            b_comp = 12 * inb + 24 * n_runs + inb * (4 + 4) * C_ * (level + 1)
            rec["compulsory_bytes_if_every_vertex_were_read"] = 12 * nv + inb * (4 + 4) * C_ * (level + 1)
            traffic = cfg4_pmc_traffic(level)
            rec["roofline"] = {"bound": "hbm", "kernel": "pair_stage_kernel (gates + residuals; + "
                               "prologue, epilogue A)", "achieved": b_comp / (t_k * 1e-6) / 1e9, "peak": 8000.0,
                               "unit": "GB/s", "frac": b_comp / (t_k * 1e-6) / 8e12,
                               "compulsory_bytes": b_comp, "traffic": traffic,
                               "no_reuse_bytes": b_alg, "frac_no_reuse": b_alg / (t_k * 1e-6) / 8e12,
                               "note": "frac: compulsory bytes (every IN-BOUND source vertex once + the box table, every "
                                       "overlapped destination row once per pair) over the stage's time; frac_no_reuse counts each of the 8 corner "
                                       "fetches of an in-bound vertex (SURVEY 8d's convention) -- most of those are L2 "
                                       "hits, which is why it can approach 1 without HBM being busy; traffic: PMC bytes "
                                       "of the pair kernel per launch (profiles/*_pmc_summary.json, cfg4_align), null "
                                       "when not measured on these kernel sources"}
            del plan
        else:
            flat = torch.zeros(6 * S + 1, device=dev)
            red = lambda: mdist.all_reduce_sum(flat)                                           # noqa: E731
            for _ in range(5):
                red()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                red()
            torch.cuda.synchronize()
            rec["all_reduce_us"] = (time.perf_counter() - t0) / 50 * 1e6
            rec["all_reduce_share"] = rec["all_reduce_us"] * 1e-6 / per_it
            # the policy of miso_amd.dist (alignment_mode): a level cheaper than its collective runs replicated
            rec["mode"] = dist_info.get("mode")
            rec["policy"] = {k: dist_info.get(k) for k in ("pair_stage_estimate_us", "all_reduce_us", "costs_agree",
                                                           "pairs_this_rank", "world")}
            rec["pairs_this_rank"] = dist_info.get("pairs_this_rank")
            # the same level on ONE rank's GPU (every rank runs the single-process loop side by side, no collective)
            one = min(run(6 * iters, one_rank=True)[1] for _ in range(2))
            rec["ms_per_iteration_one_rank"] = one * 1e3
            rec["speedup_vs_one_rank"] = one / per_it
        out[f"level{level}"] = rec
    for s in range(S):
        atlas.set_submap_pose_correction(s, *start[s])
    return out


def map_cfg3(dev, atlas, dist, steps=300, n=540000):
    """cfg-3: the S submaps mapped submap-parallel (rank r owns {s : s % world == r}, no per-step collective:
    decoder frozen, grids disjoint), `steps` GridTrainer.train_step iterations of 540 000 samples each per submap
    (300: what demo/build_submaps.py:76-91 trains a submap for), then sync_submaps (every owner broadcasts what it
    owns as one buffer).  Wall time over all ranks, broadcast included, and the two parts separately."""
    import tempfile
    import miso_amd.grid_opt.loss as L
    from miso_amd import dist as mdist
    from miso_amd.grid_opt.trainer import GridTrainer
    S = atlas.num_submaps
    mine = mdist.owned_submaps(S)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([6.0, 2.5, 6.0])
    gt = {"sdf": (torch.rand(1, n, 1, generator=g) * 0.2 - 0.1).to(dev),
          "sdf_valid": torch.ones(1, n, 1, device=dev), "sdf_signs": torch.zeros(1, n, 1, device=dev)}
    tcfg = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1, "ckpt_every": -1,
            "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": tempfile.mkdtemp(),
            "relchange_tol": 0, "max_epochs_in_level": 1000, "grid_training_mode": "joint"}
    trainers = []
    for s in mine:
        net = atlas.get_submap(s)
        net.unlock_feature()
        net.lock_pose()
        lossf = L.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
        inp = {"coords_frame": x[None].to(dev), "sample_frame_ids": torch.full((1, n, 1), s, dtype=torch.int64, device=dev),
               "weights": torch.ones(1, n, 1, device=dev)}
        trainers.append((GridTrainer(tcfg, net, lossf, None, None, dev, torch.float32), inp))
    for tr, inp in trainers:
        for _ in range(3):
            tr.train_step(inp, gt)
    B._maybe_fail("map_cfg3")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for tr, inp in trainers:
        for _ in range(steps):
            tr.train_step(inp, gt)
    torch.cuda.synchronize()
    t_map = time.perf_counter() - t0
    mdist.sync_submaps(atlas)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t_all = time.perf_counter() - t0
    bytes_b = sum(p.numel() * 4 for s in range(S) for p in list(atlas.get_submap(s).parameters())
                  + list(atlas.get_submap(s).buffers()))
    for s in mine:
        atlas.get_submap(s).lock_feature()
    del trainers
    torch.cuda.empty_cache()
    return {"submaps": S, "submaps_this_rank": len(mine), "steps_per_submap": steps, "samples_per_step": n,
            "map_ms_this_rank": t_map * 1e3, "wall_ms_incl_sync": t_all * 1e3, "sync_ms": (t_all - t_map) * 1e3,
            "broadcast_bytes": bytes_b, "point_samples_per_s": S * steps * n / t_all}


def extras_multi(dev, dist, dog):
    """Collective-bearing workloads at N > 1 ranks: cfg-3 (submap-parallel mapping + the grid broadcast) and cfg-4
    (alignment with the pair list sharded and one all-reduce per iteration).  A rank that throws inside one of them
    cannot rejoin the collectives the others are in: it raises the watchdog's flag and every rank abandons the extras
    together (the headline line is printed with the error)."""
    ex = dog.partial
    atlas = scannet_atlas(dev, 8)
    for key, fn in (("cfg3_map_8_submaps_parallel", lambda: map_cfg3(dev, atlas, dist)),
                    ("cfg4_align_8_submaps_sharded", lambda: (atlas.precompute_coordinates_for_alignment(),
                                                               align_cfg4(dev, atlas, dist=dist))[1])):
        try:
            ex[key] = fn()
        except Exception as exc:  # noqa: BLE001
            dog.raise_flag(f"{key} on rank {dog.rank}: {type(exc).__name__}: {exc}")
            time.sleep(3600)          # the watchdog thread ends this process within a fraction of a second
    return ex


def extras(step, dev):
    """Secondary figures SURVEY 8(d) asks for next to the headline: forward-only, the full trainer
    step with dense Adam, and the alignment pair (pairwise_loss_latent forward+backward)."""
    from miso_amd import ops
    from miso_amd.step import MappingStep
    import miso_amd.grid_opt.align.miso as AM
    from miso_amd.grid_opt.models.grid_atlas import GridAtlas
    ex = {}
    feats, meta, pack = step.features, step.meta, step.pack
    t = B.time_kernel(lambda: ops.sdf_fwd_raw(step.x, feats, meta, pack, False, out=step.sdf))
    ex["forward_only_point_samples_per_s"] = B.N_POINTS / (t * 1e-6)
    # the headline step with the exact fp32 decoder chains of rounds 1-5 (MISO_F_EXACT_F32) beside the default split form
    t_split = B.time_kernel(step.run, iters=100)
    with ops.exact_fp32():
        t_exact = B.time_kernel(step.run, iters=100)
    ex["exact_fp32_step"] = {"us": t_exact, "point_samples_per_s": B.N_POINTS / (t_exact * 1e-6),
                             "default_split_step_us": t_split,
                             "what": "the same step with MISO_F_EXACT_F32 (stream launches read the switch per launch)"}
    # full trainer step: bin + forward + loss + backward + dense Adam over all 19.2 M grid floats
    tr = MappingStep([f.clone() for f in feats], meta, pack, B.N_POINTS, "L1", 1.0, 0.0, 0.0, adam=dict(lr=1e-3),
                     use_graph=False)
    tr.set_batch(step.x, step.target)
    t = B.time_kernel(tr.run, iters=20, warm=3)
    ex["trainer_step_with_dense_adam"] = {"us": t, "point_samples_per_s": B.N_POINTS / (t * 1e-6),
                                          "adam_bytes": 28 * sum(f.numel() for f in feats)}
    del tr
    # alignment pair, ScanNet-shaped submaps (bound 20x10x20 m, cells 0.5 / 0.1 m, C=4), level 1:
    # 4.0 M cached voxel centres of src mapped into dst, loss + backward to both poses
    cfg = {"name": "grid_net", "spatial_dim": 3,
           "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                       "fix": True, "pretrained_model": None},
           "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 1e-2, "bound": [[-10., 10.], [-5., 5.], [-10., 10.]],
                    "base_cell_size": 0.5, "per_level_scale": 5, "n_levels": 2},
           "pose": {"optimize": False, "num_poses": 1}}
    torch.manual_seed(0)
    atlas = GridAtlas(cfg, device=dev)
    lb = torch.tensor(cfg["grid"]["bound"])
    for s_, tx in enumerate((0.0, 9.0)):
        atlas.add_submap(lb, torch.eye(3), torch.tensor([[tx], [0.3], [-0.4]]), num_poses=1)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
    atlas.to(dev)
    atlas.set_submap_pose_correction(1, torch.tensor([[0.02, -0.03, 0.01]], device=dev),
                                     torch.tensor([[0.1], [-0.05], [0.08]], device=dev))
    atlas.precompute_coordinates_for_alignment()
    nv = atlas.coordinates_for_alignment(0, 1).shape[0]

    def pair():
        atlas.zero_grad(set_to_none=True)
        (v,) = AM.pairwise_loss_latent(atlas, None, 0, 1, level=1, fdim=4, align_loss="L2", device=dev).values()
        v.backward()

    t = B.time_kernel(pair, iters=10, warm=2)
    ex["align_pair_latent_level1"] = {"vertices": nv, "us": t, "vertices_per_s": nv / (t * 1e-6)}

    # the same pair through the alignment driver (generic_align_multiple_submaps: fused pose-Adam loop on the device,
    # overlap gate, NaN guard) with the reference's own settings verbose + save_iterations: wall time per iteration
    class _DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return 0

    import miso_amd.grid_opt.align.base as AB
    latent = AM.latent_loss_for_level(atlas, 1, device=dev)

    def run(n_it):
        # every run starts from the same perturbed pose: the work per iteration follows the overlap
        atlas.set_submap_pose_correction(1, torch.tensor([[0.02, -0.03, 0.01]], device=dev),
                                         torch.tensor([[0.1], [-0.05], [0.08]], device=dev))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        AB.generic_align_multiple_submaps(atlas, _DS(), ("latent1", latent), num_iters=n_it - 1, lr=1e-3, verbose=True,
                                          save_iterations=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(12)                                  # warm-up
    t20, t120 = min(run(20), run(20)), min(run(120), run(120))
    ex["align_level1_driver"] = {"pairs": 1, "vertices": nv, "ms_20_iterations": t20 * 1e3,
                                 "ms_120_iterations": t120 * 1e3, "us_per_further_iteration": (t120 - t20) / 100 * 1e6,
                                 "settings": "verbose=True, save_iterations=True (configs/rgbd/scannet.yaml:65-66)"}
    del atlas
    torch.cuda.empty_cache()

    def cfg4():
        at = scannet_atlas(dev, 8)
        at.precompute_coordinates_for_alignment()
        return align_cfg4(dev, at)

    for key, fn in (("cfg4_align_8_submaps_28_pairs", cfg4),
                    ("sample_generation_scannet", lambda: sample_generation(dev)),
                    ("mesh_extraction_256", lambda: mesh_extraction(step, dev)),
                    ("atlas_mesh_extraction_8_submaps_512", lambda: atlas_mesh_extraction(dev)),
                    ("eikonal_step_262144pts", lambda: eikonal_step(dev)),
                    ("trainer_step_other_shapes", lambda: trainer_steps(dev)),
                    ("slam_components", lambda: slam_components(dev))):
        try:
            ex[key] = fn()
        except Exception as exc:  # noqa: BLE001
            ex[key] = {"error": f"{type(exc).__name__}: {exc}"}
    return ex


def eikonal_step(dev):
    """A second-order step at cfg-2 (grid_opt/loss_isdf.py:96-152,367-377; loss.py:638-665): sdf = fused(x), g = d sdf / d x
    with create_graph=True, loss = mean (|g| - 1)^2 + mean |sdf|, backward to the three grids.  The double backward stays in
    the library (ops._SdfFusedBackward: sdf_bwd_kernel keeps its d-feat rows, the second-order encode differentiates them);
    `torch_linear_chain_us` is the same step with the graph rebuilt from encode + torch.nn.functional.linear (rounds 1-5)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eikonal_bench as EB
    from miso_amd import ops
    args = EB.build(dev)
    feats, meta, pack, x = args

    def wall(iters=10):
        # (the first ~15 steps of either form run 10 - 60 % slower: allocator growth and the device's clock ramp)
        for _ in range(20):
            EB.step(*args)
        best = float("inf")
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                EB.step(*args)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / iters * 1e6)
        return best

    t_fused = wall()
    ops._BWD2_TORCH = True
    try:
        t_torch = wall()
    finally:
        ops._BWD2_TORCH = False
    # the library launches of the step, each alone (HIP events)
    n = x.shape[0]
    fd = [f.detach() for f in feats]
    sdf, mask = ops.sdf_fwd_raw(x, fd, meta, pack, True)
    gs = torch.ones(n, 1, device=dev)
    t_fwd = B.time_kernel(lambda: ops.sdf_fwd_raw(x, fd, meta, pack, True, out=sdf, mask=mask))
    t_rows = B.time_kernel(lambda: ops.sdf_bwd_rows_raw(x, fd, meta, pack, gs, mask, True, [False] * 3))
    _, _, rows = ops.sdf_bwd_rows_raw(x, fd, meta, pack, gs, mask, True, [False] * 3)
    ggx = torch.randn(n, 3, device=dev)
    t_bwd2 = B.time_kernel(lambda: ops.encode_bwd2_raw(x, fd, meta, rows, ggx, None, True, [True] * 3))
    return {"us": t_fused, "torch_linear_chain_us": t_torch, "speedup": t_torch / t_fused,
            "kernels_us": {"sdf_fwd_kernel(+sign bits)": t_fwd, "sdf_bwd_kernel(d sdf/d x + d-feat rows)": t_rows,
                           "encode_bwd2 (double backward: grids + x, incl. its pull)": t_bwd2},
            "note": "wall time per step incl. autograd's Python; the |sdf| term adds one ordinary fused backward"}


def atlas_mesh_extraction(dev, res=512, res_loop=192):
    """The demos' final global mesh (demo/align_submaps.py:99, full_slam_scannet.py:116: save_mesh(atlas, global bound,
    resolution=512)): the SDF volume of an 8-submap ScanNet-shaped atlas on a res^3 lattice through the fused atlas query
    (miso_atlas_sdf_fwd: one launch per slab, points generated in the kernel) next to the op-by-op per-submap loop of the
    reference's structure (grid_atlas.py:374-399; timed at res_loop^3 -- at 512^3 it runs for seconds), and marching cubes
    on the fused volume.  Bytes: the volume written once (4 B per point); the grids (8 x 16.1 M floats) stay in the caches."""
    import miso_amd.grid_opt.utils.utils_sdf as US
    from miso_amd import ops
    at = scannet_atlas(dev, 8)
    gb = at.global_bound(device="cpu").detach()
    lo, hi = gb[:, 0], gb[:, 1]

    def field(r, fused):
        q = (lambda p: at(p)) if fused else (lambda p: loop(p))
        with torch.no_grad():
            return US.extract_fields_device(lo, hi, r, q, dev, lattice_func=at.sdf_on_lattice if fused else None)

    def loop(p):
        with torch.enable_grad():        # autograd on: GridAtlas.forward runs its per-submap loop
            return at(p).detach()

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    t_loop = timed(lambda: field(res_loop, False), reps=2)
    t_fused_small = timed(lambda: field(res_loop, True))
    t_fused = timed(lambda: field(res, True))
    vol = field(res, True)
    inside = float((vol != vol.flatten()[0]).float().mean())          # (corner 0 of the bounding box lies in no submap)
    iso = float(vol[vol != vol.flatten()[0]].median())
    t_mc = timed(lambda: ops.marching_cubes(vol, iso), reps=2)
    v, f = ops.marching_cubes(vol, iso)
    n = res ** 3
    out = {"resolution": res, "submaps": 8, "points": n,
           "fused_ms": t_fused * 1e3, "fused_points_per_s": n / t_fused,
           "loop_resolution": res_loop, "loop_ms": t_loop * 1e3, "loop_points_per_s": res_loop ** 3 / t_loop,
           "fused_ms_at_loop_resolution": t_fused_small * 1e3,
           "speedup_vs_loop_same_lattice": t_loop / t_fused_small,
           "fraction_of_lattice_inside_a_submap": inside,
           "volume_write_GBps": 4 * n / t_fused / 1e9, "hbm_frac_on_volume_bytes": 4 * n / t_fused / 8e12,
           "marching_cubes_ms": t_mc * 1e3, "triangles": int(f.shape[0]), "iso": iso,
           "note": "compute-bound (one decoder evaluation per point inside any submap): 4 B per point of compulsory HBM traffic"}
    del vol, at
    torch.cuda.empty_cache()
    return out


def mesh_extraction(step, dev, res=256):
    """The step after the path: SDF volume of the cfg-2 submap on a res^3 lattice (slab-wise fused
    forward) and marching cubes on it where it lies, against the numpy oracle on the same volume (kind 'port';
    the reference's PyMCubes is not in this image)."""
    from miso_amd import ops
    from miso_amd.grid_opt.utils import utils_sdf as US
    from oracle import mcubes_ref as M  # checker / baseline leg only
    feats, meta, pack = step.features, step.meta, step.pack
    lo, hi = torch.tensor(meta.bound_min), torch.tensor(meta.bound_max)
    query = lambda p: ops.sdf_fwd_raw(p, feats, meta, pack, False)[0]
    t0 = time.perf_counter()
    vol = US.extract_fields_device(lo, hi, res, query, device=dev)
    torch.cuda.synchronize()
    t_first = (time.perf_counter() - t0) * 1e6        # the first call of a process: allocations, first launches
    t0 = time.perf_counter()
    vol = US.extract_fields_device(lo, hi, res, query, device=dev)
    torch.cuda.synchronize()
    t_field = (time.perf_counter() - t0) * 1e6
    iso = float(vol.median())                 # a random decoder's field need not cross zero
    t_mc = B.time_kernel(lambda: ops.marching_cubes(vol, iso), iters=10, warm=2)
    v, f = ops.marching_cubes(vol, iso)
    t0 = time.perf_counter()
    rv, rf = M.marching_cubes(vol.cpu().numpy(), iso)
    t_cpu = (time.perf_counter() - t0) * 1e6
    same = bool((f.cpu().numpy() == rf).all() and (v.cpu().numpy() == rv).all())
    return {"resolution": res, "field_us": t_field, "field_us_first_call": t_first,
            "field_points_per_s": res ** 3 / (t_field * 1e-6), "marching_cubes_us": t_mc, "triangles": int(f.shape[0]),
            "vertices": int(v.shape[0]), "volume_GBps": 4 * res ** 3 / t_mc / 1e3, "cpu_port_us": t_cpu,
            "equals_cpu_port": same}


def trainer_steps(dev):
    """GridTrainer.train_step (keyframe transform -> captured step -> Adam) at the grid shapes of BASELINE configs 3
    and 5, wall time per step: a ScanNet submap (20x10x20 m, cells 0.5/0.1 m, C=4; 540 000 samples around the
    middle of the bound) and a Newer College submap (120x120x20 m, cells 1.0/0.2 m: 144 M floats in the fine level;
    6 144 samples around the sensor)."""
    import tempfile
    import miso_amd.grid_opt.loss as L
    from miso_amd.grid_opt.models.grid_net import GridNet
    from miso_amd.grid_opt.trainer import GridTrainer
    out = {}
    shapes = {"cfg3_scannet_540000pts": ([[-10., 10.], [-5., 5.], [-10., 10.]], 0.5, 540000, [6.0, 2.5, 6.0], [0., 0., 0.],
                                         (4, 5, 2)),
              "cfg5_newer_college_6144pts": ([[-60., 60.], [-60., 60.], [-5., 15.]], 1.0, 6144, [25.0, 25.0, 4.0],
                                             [5.0, -8.0, 2.0], (4, 5, 2)),
              # the headline grid through the same entry point (GridTrainer.train_step incl. dense Adam over 19.2 M floats)
              "cfg2_262144pts": ([[-1., 1.]] * 3, 2.0 / 32, 262144, [1.0, 1.0, 1.0], [0., 0., 0.], (8, 2, 3))}
    for name, (bound, cell, n, half, mid, (fdim, scale, n_levels)) in shapes.items():
        cfg = {"name": "grid_net", "spatial_dim": 3,
               "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                           "fix": True, "pretrained_model": None},
               "grid": {"type": "regular", "feature_dim": fdim, "init_stddev": 1e-2, "bound": bound,
                        "base_cell_size": cell, "per_level_scale": scale, "n_levels": n_levels},
               "pose": {"optimize": False, "num_poses": 1}}
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor(half) + torch.tensor(mid)
        batch = ({"coords_frame": x[None].to(dev), "sample_frame_ids": torch.zeros(1, n, 1, dtype=torch.int64, device=dev),
                  "weights": torch.ones(1, n, 1, device=dev)},
                 {"sdf": (torch.rand(1, n, 1, generator=g) * 0.2 - 0.1).to(dev),
                  "sdf_valid": torch.ones(1, n, 1, device=dev), "sdf_signs": torch.zeros(1, n, 1, device=dev)})
        torch.manual_seed(0)
        net = GridNet(cfg, device=dev).to(dev)
        net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
        net.unlock_feature()
        net.lock_pose()
        tcfg = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1, "ckpt_every": -1,
                "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": tempfile.mkdtemp(),
                "relchange_tol": 0, "max_epochs_in_level": 1000, "grid_training_mode": "joint"}
        lossf = L.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
        tr = GridTrainer(tcfg, net, lossf, None, None, dev, torch.float32)
        tw, k = time.perf_counter(), 0
        while k < 5 or time.perf_counter() - tw < 0.03:       # 30 ms of steps: past the device's clock ramp (B.settle_device)
            tr.train_step(*batch)
            k += 1
        torch.cuda.synchronize()
        import gc
        gc.disable()
        us = float("inf")
        for _ in range(3):                          # best of three loops: a one-off host hiccup once read 345 for 305 us
            t0 = time.perf_counter()
            for _ in range(60):
                tr.train_step(*batch)
            torch.cuda.synchronize()
            us = min(us, (time.perf_counter() - t0) / 60 * 1e6)
        gc.enable()
        # roofline of the step at this shape: SURVEY 8(d)'s algorithmic bytes per point-sample over the
        # whole step, the launches of the mapping step alone (HIP events), and -- where a committed counter summary of
        # tools/pmc_trainer.sh exists -- the HBM traffic and the fp32 atomic requests of the dominant kernel
        L_, C_ = n_levels, fdim
        b_alg = 20 + 64 * L_ * C_
        roof = {"bound": "hbm", "algorithmic_bytes_per_point": b_alg, "achieved": n * b_alg / (us * 1e-6) / 1e9,
                "peak": 8000.0, "unit": "GB/s", "frac": n * b_alg / (us * 1e-6) / 8e12, "traffic": None}
        plan = tr.__dict__.get("_fast_plan")
        if plan is not None:
            roof["mapping_launches_us"] = B.time_kernel(plan.step._launch, iters=30)
        pmc_file = os.path.join(ROOT, "profiles", "r06_pmc_trainer_%s.json" % ("scannet" if "scannet" in name else "ncd"))
        if "cfg2" not in name and os.path.exists(pmc_file):
            js = json.load(open(pmc_file))
            dom_k = max((k for k in js if k.startswith("sdf_train_kernel")), key=lambda k: js[k].get("avg_us", 0), default=None)
            if dom_k:
                d = js[dom_k]
                req = d.get("TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum", 0.0)
                roof.update({"dominant_kernel": dom_k, "dominant_kernel_us_profiled": d.get("avg_us"),
                             "traffic": (2 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024,
                             "traffic_source": os.path.basename(pmc_file) + " (FETCH_SIZE x 2 + WRITE_SIZE, KiB; gfx950)",
                             "fp32_atomic_requests_per_launch": req,
                             "atomics_executed_memory_side": d.get("TCC_EA0_ATOMIC_sum"),
                             "atomic_floor_us_at_21G_requests_per_s": req / 21e9 * 1e6,
                             "binding_resource": "memory-side fp32 atomic units: every L2 atomic request leaves for the fabric "
                                                 "(TCC_EA0_ATOMIC == TCC_ATOMIC == TCP->TCC requests), ~21 G requests/s chip-wide "
                                                 "(tools/ubench/atomics.hip)" if req > 1e5 else "latency (one chunk per wavefront)"})
        out[name] = {"us_per_step": us, "grid_floats": sum(f.feature.numel() for f in net.features),
                     "point_samples_per_s": n / (us * 1e-6), "roofline": roof,
                     "path": ("captured step + optimizer.step()" if tr.__dict__.get("_fast_plan") is None else
                              "one graph replay incl. Adam (_FastMappingPlan)" if tr._fast_plan.step._use_graph else
                              "stream launches incl. Adam (_FastMappingPlan)")}
        del tr, net
        torch.cuda.empty_cache()
    return out


def slam_components(dev):
    """The per-frame pieces of the SLAM loop as the reference's drivers call them (wall time, host included):
    Mapper.mapping -- a NEW GridTrainer per call, coordinate+joint schedule, 10 iterations (slam/mapper.py:65-97) -- at
    the ScanNet shape, and one tracker iteration with either solver (slam/tracker.py: lm_step / track_window) at 16 384
    samples."""
    import tempfile
    from miso_amd.grid_opt.models.grid_net import GridNet
    from miso_amd.grid_opt.slam.mapper import Mapper
    from miso_amd.grid_opt.slam.tracker import Tracker
    out = {}
    cfg_m = {"name": "grid_net", "spatial_dim": 3,
             "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                         "fix": True, "pretrained_model": None},
             "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 1e-2, "bound": [[-10., 10.], [-5., 5.], [-10., 10.]],
                      "base_cell_size": 0.5, "per_level_scale": 5, "n_levels": 2},
             "pose": {"optimize": True, "num_poses": 4}}

    def dataset(n, frame):
        g = torch.Generator().manual_seed(1)
        pts = ((torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([6.0, 2.5, 6.0])).to(dev)
        sdf = (torch.rand(n, 1, generator=g) * 0.2 - 0.1).to(dev)
        ids = (torch.randint(0, 4, (n, 1), generator=g) if frame is None else torch.full((n, 1), frame)).to(dev)
        one, zero = torch.ones(n, 1, device=dev), torch.zeros(n, 1, device=dev)

        class DS(torch.utils.data.Dataset):
            def select_keyframes(self, kfs):
                pass

            def __len__(self):
                return 1

            def __getitem__(self, i):
                return ({"coords_frame": pts, "sample_frame_ids": ids, "weights": one},
                        {"sdf": sdf, "sdf_valid": one, "sdf_signs": zero})
        return DS()

    def net():
        torch.manual_seed(0)
        m = GridNet(cfg_m, device=dev).to(dev)
        for k in range(4):
            m.set_initial_kf_pose(k, torch.eye(3), torch.tensor([[0.05 * k], [0.0], [0.02 * k]]), kf_key=f"KF{k}")
        return m

    log = tempfile.mkdtemp()
    train = {"trainer": "base", "verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 50,
             "ckpt_every": -1, "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": log,
             "relchange_tol": 0, "max_epochs_in_level": 100, "grid_training_mode": "coordinate+joint"}
    mapping = {"learning_rate": 1e-3, "loss_type": "L1", "weight_sdf": 1.0, "weight_eik": 0.0, "weight_fs": 0.1,
               "trunc_dist": 0.15, "finite_diff_eps": 0.01, "grad_method": "finitediff", "eik_trunc_dist": 0.024,
               "verbose": False}
    mp = Mapper(net(), dataset(540000, None), {"device": dev, "train": train, "mapping": mapping})
    for _ in range(2):
        mp.mapping([0, 1, 2, 3], iterations=10, level_iterations=5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        mp.mapping([0, 1, 2, 3], iterations=10, level_iterations=5)
    torch.cuda.synchronize()
    out["mapper_mapping_540000pts_10_iterations"] = {"ms_per_call": (time.perf_counter() - t0) / 5 * 1e3,
                                                     "schedule": "coordinate+joint, 5 iterations per level, new GridTrainer per call"}
    del mp
    torch.cuda.empty_cache()
    for solver, lt in (("lm", "GM"), ("adam", "L1")):
        tracking = {"learning_rate": 1e-3, "verbose": False, "gm_scale_sdf": 0.1, "lm_lambda": 1e-4, "lm_max_iter": 10,
                    "lm_tol_deg": 0.0, "lm_tol_m": 0.0, "loss_type": lt, "trunc_dist": None, "solver": solver}
        trk = Tracker(net(), dataset(16384, 1), {"device": dev, "train": train, "tracking": tracking})
        if solver == "lm":
            for _ in range(3):
                trk.lm_step(1)
            best = float("inf")
            for _ in range(2):                      # best of two loops: a one-off host hiccup once cost 40 ms here
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    trk.lm_step(1)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 30 * 1e6)
            out["tracker_lm_step_16384pts"] = {"us_per_step": best}
        else:
            trk.track_window([1], iterations=15)
            best = float("inf")
            for _ in range(2):                      # (best of two loops, as above)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    trk.track_window([1], iterations=15)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 60 * 1e6)
            out["tracker_adam_window_16384pts"] = {"us_per_iteration": best, "iterations_per_window": 15}
        del trk
    torch.cuda.empty_cache()
    return out


def sample_generation(dev):
    """one PosedSdfRgbd.__getitem__ at the ScanNet knobs (configs/rgbd/scannet.yaml:107-111 --
    100 keyframes x 200 rays x (8 + 19) samples = 540 000 rows): miso_sample_rays alone, the dataset call
    (draws + sampler + row count read-back), and the CPU restatement of the reference on the same draws."""
    from miso_amd import ops
    from miso_amd.grid_opt.datasets.sdf_rgbd import PosedSdfRgbd
    from miso_amd.grid_opt.utils.utils_data import CameraParameters
    from oracle import ref_torch as R
    NF, H, W, rays, n_strat, n_surf = 100, 480, 640, 200, 19, 8
    g = torch.Generator().manual_seed(3)
    depth = torch.rand(NF, H, W, generator=g) * 4.0 + 0.5
    depth[torch.rand(NF, H, W, generator=g) < 0.1] = 0.0
    ang = torch.rand(NF, generator=g) * 6.28
    Rm = torch.eye(3).repeat(NF, 1, 1)
    Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 2, 0], Rm[:, 2, 2] = ang.cos(), ang.sin(), -ang.sin(), ang.cos()
    t = torch.rand(NF, 3, 1, generator=g) * 10 - 5
    cp = CameraParameters(fx=577.6, fy=578.7, cx=318.9, cy=242.7, H=H, W=W)
    normals = torch.ones(NF, H, W, 3)          # estimation is one-time set-up, not part of the per-iteration cost
    ds = PosedSdfRgbd.from_frames(depth, Rm, t, cp, n_rays=rays, n_strat_samples=n_strat, n_surf_samples=n_surf,
                                  trunc_dist=0.15, device=dev, normals=normals)
    n = NF * rays
    ph = torch.randint(0, H, (n,), generator=g)
    pw = torch.randint(0, W, (n,), generator=g)
    u = torch.rand(n, n_strat, generator=g)
    gg = torch.randn(n, n_surf - 1, generator=g) * 0.1
    draws = tuple(a.to(dev) for a in (ph, pw, u, gg))
    out = ops.RayBatch(n, n_strat + n_surf, dev)
    t_kernel = B.time_kernel(lambda: ds.sample_batch(out=out, draws=draws), iters=20, warm=3)
    rows = out.rows()
    for _ in range(3):
        ds[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ds[0]
    torch.cuda.synchronize()
    t_item = (time.perf_counter() - t0) / 10 * 1e6
    pb = torch.arange(NF).repeat_interleave(rays)
    knobs = dict(min_depth=0.07, dist_behind_surf=0.1, trunc_dist=0.15, n_strat=n_strat, n_surf=n_surf)
    Tm = torch.eye(4).repeat(NF, 1, 1)
    Tm[:, :3, :3], Tm[:, :3, 3:] = Rm, t
    t0 = time.perf_counter()
    for _ in range(3):
        R.rgbd_sdf_samples(ds._depth_batch.cpu(), Tm, Rm, t, (cp.fx, cp.fy, cp.cx, cp.cy), pb, ph, pw, u, gg,
                           normals=normals, **knobs)
    t_cpu = (time.perf_counter() - t0) / 3 * 1e6
    # algorithmic bytes per row: coords 12 + frame id 8 + labels 16 written; per ray 4 depth + 16 pixel + 4 per draw read
    b_alg = rows * 36 + n * (20 + 4 * (n_strat + n_surf - 1))
    return {"rows": rows, "sampler_us": t_kernel, "rows_per_s": rows / (t_kernel * 1e-6),
            "algorithmic_GBps": b_alg / (t_kernel * 1e-6) / 1e9, "dataset_getitem_us": t_item,
            "cpu_port_us": t_cpu, "cpu_port_note": "oracle.rgbd_sdf_samples (vectorised: the reference's per-keyframe "
                                                   "Python loop of sdf_rgbd.py:438-445 is not in it)"}
