"""Headline benchmark: 3-D point-samples/s, encode+decode forward+backward.

Workload (BASELINE.json configs[1], "cfg-2"): one submap per GPU, 3-level
{32,64,128}^3 feature grid (C=8, channels-last), frozen decoder MLP 24-64-64-1
(seeded random weights stand in for the unavailable decoder_indoor.pt), one
batch of 262 144 uniform-in-bbox points per GPU per step, L1 regression loss,
gradients to every grid level (decoder frozen, configs/rgbd/scannet.yaml:16).
A step = bin the batch by spatial tile -> fused forward -> loss -> fused backward
(MFMA pass + owner-computes gradient pull, which needs no zero-fill), exactly K
times inside the timed region (inputs resident in HBM).  Submaps are independent, so
N GPUs run N submaps with no data-path collective (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the contract in the task description).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 262144
LEVELS = (32, 64, 128)
C = 8
HIDDEN = 64


def build_workload(dev, rank):
    from miso_amd import ops
    from miso_amd.step import MappingStep
    g = torch.Generator().manual_seed(0)
    feats = [(torch.randn(1, C, s, s, s, generator=g) * 1e-2).to(dev).contiguous(
        memory_format=torch.channels_last_3d) for s in LEVELS]
    torch.manual_seed(0)
    lin = [torch.nn.Linear(C * len(LEVELS), HIDDEN), torch.nn.Linear(HIDDEN, HIDDEN), torch.nn.Linear(HIDDEN, 1)]
    ws = [l.weight.detach().to(dev) for l in lin]
    bs = [l.bias.detach().to(dev) for l in lin]
    meta = ops.GridMeta.from_bound([[-1.0, 1.0]] * 3)
    pack = ops.DecoderPack(ws, bs)
    gp = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(N_POINTS, 3, generator=gp) * 2 - 1
    targ = torch.randn(N_POINTS, 1, generator=gp) * 0.1
    # keep_sdf=False: a training step needs the loss and the gradients, not the per-point SDF in the
    # caller's order (the accuracy check below evaluates the forward separately)
    # MISO_BENCH_LAUNCH=graph|stream (dev: A/B of the two launch modes; default: MappingStep's own choice by batch size)
    mode = {"graph": True, "stream": False}.get(os.environ.get("MISO_BENCH_LAUNCH", ""), None)
    step = MappingStep(feats, meta, pack, N_POINTS, loss_type="L1", weight_sdf=1.0, weight_fs=0.0, keep_sdf=False,
                       use_graph=mode)
    step.set_batch(x.to(dev), targ.to(dev))
    return step, (feats, ws, bs, x, targ)


SETTLE_STEPS = 256


def settle_device(step):
    """Untimed steps in front of the W warmup steps, so that the timed region measures the sustained rate: an MI355X
    that has been without work for more than ~2 ms lowers its clocks and needs ~70 steps (11 ms) of this workload to
    bring them back (tools/ramp_probe.py, us per step in blocks of 5 from a cold device: 166 166 166 164 164 164 163 161
    162 160 162 160 159 158 158 -> 156-157 sustained).  The driver's default run times 20 steps = 3.3 ms after 5 warmup
    steps, i.e. entirely inside that ramp (172 us per step, either launch mode).  Reported as `settle_steps`."""
    for _ in range(SETTLE_STEPS):
        step.run()


def time_kernel(fn, iters=30, warm=5, warm_ms=12.0):
    """Average duration (us) of `fn` (one launch) with HIP events on the launch stream, after `warm_ms` of the same
    launches (clock ramp, see settle_device) with no idle gap in front of the timed ones."""
    t0 = time.perf_counter()
    n = 0
    while n < warm or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        n += 1
        if n % 32 == 0:
            torch.cuda.synchronize()        # the host must not run a thousand launches ahead of a 100 us kernel
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def host_cpu():
    """CPU model string and physical core count of this host (lscpu), for the cpu_baseline record."""
    import subprocess
    info = {}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = dict((ln.split(":", 1)[0].strip(), ln.split(":", 1)[1].strip()) for ln in txt.splitlines() if ":" in ln)
        info["cpu_model"] = kv.get("Model name")
        sockets, cps = int(kv.get("Socket(s)", 0) or 0), int(kv.get("Core(s) per socket", 0) or 0)
        info["physical_cores"] = sockets * cps or None
        info["logical_cpus"] = int(kv.get("CPU(s)", 0) or 0) or None
    except Exception as exc:  # noqa: BLE001
        info["cpu_model"] = f"unknown ({type(exc).__name__})"
    return info


def cpu_baseline(data, budget_s=20.0):
    """The reference's op sequence (per-level F.grid_sample -> cat -> nn.Sequential -> L1 ->
    backward) restated with stock torch CPU ops (oracle/ref_torch.py, kind 'port'),
    timed on this host on a bounded sample of whole 262 144-point iterations."""
    from oracle import ref_torch as R  # checker / baseline leg only
    feats, ws, bs, x, targ = data
    fc = [f.detach().cpu().contiguous().requires_grad_(True) for f in feats]
    wc, bc = [w.cpu() for w in ws], [b.cpu() for b in bs]
    bound = torch.tensor([[-1.0, 1.0]] * 3)
    cores = torch.get_num_threads()

    def it():
        for f in fc:
            f.grad = None
        pred = R.sdf_stock(fc, bound, x, wc, bc)
        loss = R.miso_loss_regression(pred, targ, None, None, "L1")
        loss.backward()
        return pred

    it()  # warm-up
    t0 = time.perf_counter()
    k = 0
    while True:
        pred = it()
        k += 1
        if time.perf_counter() - t0 > budget_s or k >= 12:
            break
    dt = (time.perf_counter() - t0) / k
    # the same on ONE thread: ATen's 3-D grid_sample is serial for batch size 1, so the figure
    # barely moves with the core count
    torch.set_num_threads(1)
    try:
        t1 = time.perf_counter()
        it()
        it()
        dt1 = (time.perf_counter() - t1) / 2
    finally:
        torch.set_num_threads(cores)
    # the baseline is the FASTER of the two (on a 128-thread host the one-thread run wins: thread launch and
    # reduction overheads around a serial grid_sample); both are reported
    v_all, v_one = N_POINTS / dt, N_POINTS / dt1
    return {"value": max(v_all, v_one), "unit": "point-samples/s", "cores": cores if v_all >= v_one else 1,
            "kind": "port", **host_cpu(),
            "sample": f"{k} full fwd+bwd iterations of 262144 points on {cores} threads (stock torch CPU ops arranged "
                      f"as the reference: F.grid_sample per level, cat, nn.Sequential, L1), {dt:.2f} s each, and 2 on "
                      f"one thread, {dt1:.2f} s each; value = the faster",
            "all_threads_value": v_all, "all_threads": cores, "one_thread_value": v_one}, pred


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
            t_k = time_kernel(plan.iteration_a, iters=10, warm=2)
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
    _maybe_fail("map_cfg3")
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


def _maybe_fail(where):
    """dev / test: MISO_BENCH_FAIL=<rank>:<where> makes that rank throw there (tests the watchdog below)."""
    spec = os.environ.get("MISO_BENCH_FAIL")
    if spec and spec == f"{os.environ.get('RANK', '0')}:{where}":
        raise RuntimeError(f"injected failure in {where}")


class Watchdog:
    """Keeps the headline line safe from the collective-bearing extras.  One thread per rank: when the shared wall-clock
    deadline passes, or ANY rank has raised the abort flag (a file named after the rendezvous: one node, so every rank
    sees it), rank 0 prints the finished headline record with extras_multi_gpu = {"error": ...}, writes `<flag>.done`
    and leaves; the other ranks wait for that file (a few seconds at most) before they leave, and leave with status 0
    when they saw it -- a launcher that ends every rank at the first non-zero status must not get one before rank 0
    has printed.  A rank blocked in an all-reduce whose peer threw would otherwise sit there until the process-group
    timeout.  (Exiting is all it does: no re-exec of a GPU process.)

    Lifecycle of the flag: its name carries the rendezvous port AND a per-run token (the launcher's, or rank 0's start
    time broadcast at construction), rank 0 removes leftovers at CONSTRUCTION -- which callers place right after the
    process group is up and before a barrier, i.e. before any rank can raise it -- and again when it expires or stops."""

    PEER_GRACE_S = 5.0

    def __init__(self, rank, budget_s, dist=None):
        import threading
        self.rank = rank
        t0 = time.time()
        token = os.environ.get("MISO_BENCH_TOKEN") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
        if dist is not None and dist.is_initialized():
            import torch
            t = torch.tensor([t0], dtype=torch.float64)
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = t.to(dev)
            dist.broadcast(t, src=0)              # one clock for the deadline, one token for the flag's name
            t0 = float(t.item())
            token = f"{token}_{int(t0 * 1e3)}"
        # rank 0 expires at the deadline, the peers a grace period later: its print comes first
        self.deadline = t0 + budget_s + (0.0 if rank == 0 else self.PEER_GRACE_S)
        self.flag = os.path.join("/tmp", f"miso_bench_abort_{os.environ.get('MASTER_PORT', '0')}_{token}")
        self.done = self.flag + ".done"
        self.headline = None          # rank 0: the finished record
        self.partial = {}             # extras that did finish
        self._off = threading.Event()
        if rank == 0:
            self._cleanup()
        self._th = threading.Thread(target=self._watch, daemon=True)

    def _cleanup(self):
        for f in (self.flag, self.done):
            try:
                os.remove(f)
            except OSError:
                pass

    def start(self):
        self._th.start()

    def raise_flag(self, reason):
        try:
            with open(self.flag, "x") as f:         # the FIRST reason stays (a peer that dies makes the others fail too)
                f.write(reason)
        except OSError:
            pass

    def _watch(self):
        while not self._off.wait(0.25):
            reason = None
            if os.path.exists(self.flag):
                try:
                    reason = open(self.flag).read() or "a rank failed"
                except OSError:
                    reason = "a rank failed"
            elif self.rank != 0 and os.path.exists(self.done):
                reason = "rank 0 has printed the record and left"       # (it removes the flag as it goes)
            elif time.time() > self.deadline:
                reason = "wall-clock budget of the multi-GPU extras exceeded"
            if reason is not None:
                self.expire(reason)

    def expire(self, reason):
        if self.rank == 0:
            if self.headline is not None:
                out = dict(self.headline)
                out["extras_multi_gpu"] = dict(self.partial, error=reason)
                sys.stdout.write(json.dumps(out) + "\n")
                sys.stdout.flush()
            try:
                open(self.done, "w").close()
            except OSError:
                pass
            try:
                os.remove(self.flag)
            except OSError:
                pass
            os._exit(0)
        # a peer: let rank 0 print first
        t_end = time.time() + self.PEER_GRACE_S
        while time.time() < t_end and not os.path.exists(self.done):
            time.sleep(0.05)
        os._exit(0 if os.path.exists(self.done) else 3)

    def stop(self):
        self._off.set()
        if self.rank == 0:
            self._cleanup()


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
    t = time_kernel(lambda: ops.sdf_fwd_raw(step.x, feats, meta, pack, False, out=step.sdf))
    ex["forward_only_point_samples_per_s"] = N_POINTS / (t * 1e-6)
    # the headline step with the exact fp32 decoder chains of rounds 1-5 (MISO_F_EXACT_F32) beside the default split form
    t_split = time_kernel(step.run, iters=100)
    with ops.exact_fp32():
        t_exact = time_kernel(step.run, iters=100)
    ex["exact_fp32_step"] = {"us": t_exact, "point_samples_per_s": N_POINTS / (t_exact * 1e-6),
                             "default_split_step_us": t_split,
                             "what": "the same step with MISO_F_EXACT_F32 (stream launches read the switch per launch)"}
    # full trainer step: bin + forward + loss + backward + dense Adam over all 19.2 M grid floats
    tr = MappingStep([f.clone() for f in feats], meta, pack, N_POINTS, "L1", 1.0, 0.0, 0.0, adam=dict(lr=1e-3),
                     use_graph=False)
    tr.set_batch(step.x, step.target)
    t = time_kernel(tr.run, iters=20, warm=3)
    ex["trainer_step_with_dense_adam"] = {"us": t, "point_samples_per_s": N_POINTS / (t * 1e-6),
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

    t = time_kernel(pair, iters=10, warm=2)
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
        for _ in range(3):
            EB.step(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            EB.step(*args)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

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
    t_fwd = time_kernel(lambda: ops.sdf_fwd_raw(x, fd, meta, pack, True, out=sdf, mask=mask))
    t_rows = time_kernel(lambda: ops.sdf_bwd_rows_raw(x, fd, meta, pack, gs, mask, True, [False] * 3))
    _, _, rows = ops.sdf_bwd_rows_raw(x, fd, meta, pack, gs, mask, True, [False] * 3)
    ggx = torch.randn(n, 3, device=dev)
    t_bwd2 = time_kernel(lambda: ops.encode_bwd2_raw(x, fd, meta, rows, ggx, None, True, [True] * 3))
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
    t_field = (time.perf_counter() - t0) * 1e6
    iso = float(vol.median())                 # a random decoder's field need not cross zero
    t_mc = time_kernel(lambda: ops.marching_cubes(vol, iso), iters=10, warm=2)
    v, f = ops.marching_cubes(vol, iso)
    t0 = time.perf_counter()
    rv, rf = M.marching_cubes(vol.cpu().numpy(), iso)
    t_cpu = (time.perf_counter() - t0) * 1e6
    same = bool((f.cpu().numpy() == rf).all() and (v.cpu().numpy() == rv).all())
    return {"resolution": res, "field_us": t_field, "marching_cubes_us": t_mc, "triangles": int(f.shape[0]),
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
        while k < 5 or time.perf_counter() - tw < 0.03:       # 30 ms of steps: past the device's clock ramp (settle_device)
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
            roof["mapping_launches_us"] = time_kernel(plan.step._launch, iters=30)
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
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                trk.track_window([1], iterations=15)
            torch.cuda.synchronize()
            out["tracker_adam_window_16384pts"] = {"us_per_iteration": (time.perf_counter() - t0) / 60 * 1e6,
                                                   "iterations_per_window": 15}
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
    B, H, W, rays, n_strat, n_surf = 100, 480, 640, 200, 19, 8
    g = torch.Generator().manual_seed(3)
    depth = torch.rand(B, H, W, generator=g) * 4.0 + 0.5
    depth[torch.rand(B, H, W, generator=g) < 0.1] = 0.0
    ang = torch.rand(B, generator=g) * 6.28
    Rm = torch.eye(3).repeat(B, 1, 1)
    Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 2, 0], Rm[:, 2, 2] = ang.cos(), ang.sin(), -ang.sin(), ang.cos()
    t = torch.rand(B, 3, 1, generator=g) * 10 - 5
    cp = CameraParameters(fx=577.6, fy=578.7, cx=318.9, cy=242.7, H=H, W=W)
    normals = torch.ones(B, H, W, 3)          # estimation is one-time set-up, not part of the per-iteration cost
    ds = PosedSdfRgbd.from_frames(depth, Rm, t, cp, n_rays=rays, n_strat_samples=n_strat, n_surf_samples=n_surf,
                                  trunc_dist=0.15, device=dev, normals=normals)
    n = B * rays
    ph = torch.randint(0, H, (n,), generator=g)
    pw = torch.randint(0, W, (n,), generator=g)
    u = torch.rand(n, n_strat, generator=g)
    gg = torch.randn(n, n_surf - 1, generator=g) * 0.1
    draws = tuple(a.to(dev) for a in (ph, pw, u, gg))
    out = ops.RayBatch(n, n_strat + n_surf, dev)
    t_kernel = time_kernel(lambda: ds.sample_batch(out=out, draws=draws), iters=20, warm=3)
    rows = out.rows()
    for _ in range(3):
        ds[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ds[0]
    torch.cuda.synchronize()
    t_item = (time.perf_counter() - t0) / 10 * 1e6
    pb = torch.arange(B).repeat_interleave(rays)
    knobs = dict(min_depth=0.07, dist_behind_surf=0.1, trunc_dist=0.15, n_strat=n_strat, n_surf=n_surf)
    Tm = torch.eye(4).repeat(B, 1, 1)
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


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process makes NO GPU call (no torch.cuda.* at all); it
    starts one child per GPU with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), forwards
    rank 0's stdout (the ONE JSON line) and exits with the first non-zero child status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    token = f"{os.getpid()}_{int(time.time() * 1e3)}"         # a per-run name for the watchdog's flag file
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MISO_BENCH_TOKEN=token,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                # a dead rank leaves the others waiting in a collective: end them (exact PIDs) -- rank 0 last and only
                # after a grace period: its watchdog prints the headline record within a second of a peer's failure
                t_end = time.time() + 10.0
                while procs[0] in live and procs[0].poll() is None and time.time() < t_end:
                    time.sleep(0.1)
                for other in live:
                    if other.poll() is None:
                        other.terminate()
        time.sleep(0.2)
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # dev-only overrides to exercise the multi-rank path on a 1-GPU box (ranks share the device,
    # rendezvous over gloo); the driver's runs use one GPU per rank and RCCL
    backend = os.environ.get("MISO_BENCH_BACKEND", "nccl")
    if "MISO_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MISO_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    collective = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=300)      # a collective nobody answers ends the job in minutes, not half an hour
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        # sanity value of the collective layer: every rank contributes rank + 1 -> world (world + 1) / 2
        from miso_amd import dist as mdist
        chk = torch.full((1,), float(rank + 1), device=dev)
        mdist.all_reduce_sum(chk)
        collective = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                      "all_reduce_check": chk.item(), "all_reduce_expected": world * (world + 1) / 2}

    from miso_amd import ops
    step, data = build_workload(dev, rank)

    # the step runs as plain stream launches (MappingStep.STREAM_MIN_POINTS): the host has to stay in front of the
    # device, and a full collection of Python's collector over what `import torch` left behind is a 45 ms pause
    import gc
    gc.collect()
    gc.freeze()
    # for the record, the protocol WITHOUT the settle phase first (rank-local, no barrier): W warm-up steps and K timed
    # steps on a device that has been idle -- the figure the clock ramp produces (settle_device), reported next to `value`
    for _ in range(args.warmup):
        step.run()
    torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run()
    torch.cuda.synchronize()
    cold_elapsed = time.perf_counter() - t0
    gc.enable()
    settle_device(step)
    for _ in range(args.warmup):
        step.run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gc.disable()                      # as timeit does: no collector pause inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # rank 0 finishes the headline record FIRST (per-kernel timings, roofline: single-rank work); the collective-bearing
    # extras run after it, under a watchdog that prints the record if they hang or a rank fails
    out = headline_record(args, step, dev, world, elapsed) if rank == 0 else None
    if rank == 0:
        out["without_settle_phase"] = {
            "ms_per_step": cold_elapsed / args.steps * 1e3, "point_samples_per_s_this_rank": N_POINTS * args.steps / cold_elapsed,
            "note": "the same W warm-up + K timed steps on rank 0 right after set-up, device clocks still ramping "
                    "(DESIGN section 5); `value` is measured after settle_steps more untimed steps"}
    if rank == 0 and collective is not None:
        out.update(collective)
    multi = None
    if dist is not None and not args.no_extras:
        dog = Watchdog(rank, float(os.environ.get("MISO_BENCH_EXTRAS_BUDGET_S", "150")), dist)
        dist.barrier()                # rank 0 has cleared leftovers of the flag: from here on a rank may raise it
        dog.headline = out
        dog.start()
        multi = extras_multi(dev, dist, dog)
        dog.stop()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    if multi is not None:
        out["extras_multi_gpu"] = multi
    if world == 1 and not args.no_extras:
        try:                      # secondary figures must never cost the headline line
            out["extras"] = extras(step, dev)
        except Exception as exc:  # noqa: BLE001
            out["extras"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not args.no_cpu_baseline:
        try:
            cb, pred_cpu = cpu_baseline(data)
            out["cpu_baseline"] = cb
            sdf_gpu, _ = ops.sdf_fwd_raw(step.x, step.features, step.meta, step.pack, False)
            err = (sdf_gpu.detach().cpu() - pred_cpu.detach()).abs()
            out["sdf_L1_vs_cpu"] = {"mean": err.mean().item(), "max": err.max().item()}
            out["speedup_vs_cpu"] = out["value"] / cb["value"]
        except Exception as exc:  # noqa: BLE001  (the headline line is printed regardless)
            out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def headline_record(args, step, dev, world, elapsed):
    """The headline JSON record (rank 0): value, per-kernel durations measured live with HIP events, roofline."""
    from miso_amd import ops
    ms_per_step = elapsed / args.steps * 1e3
    value = world * N_POINTS * args.steps / elapsed

    # ---- per-kernel durations (HIP events on the launch stream) and roofline --------------------
    L = len(LEVELS)
    feats, meta, pack = step.features, step.meta, step.pack
    sb = step.sorted
    mask = torch.empty(((N_POINTS + 63) // 64) * 64 * ops.sdf_mask_words(pack), device=dev, dtype=torch.int32)
    t_sort = time_kernel(lambda: sb.sort(step.x, meta)) if sb is not None else 0.0
    fused = sb is not None and step._fused_train()
    sb_step = sb                      # what the step itself runs on: no perm[] when the step is the fused one
    if sb is not None and sb.perm is None:
        # the step's own batch carries no perm[] (the fused step does not need one); the two-launch reference below does
        sb = ops.SortedBatch(N_POINTS, dev, tiles=step.tiles).sort(step.x, meta)
    t_train = None
    if sb is not None:
        # the two-launch form of forward and backward, each timed alone (the generic entry points; what runs when a
        # level is still scattered from the backward) ...
        t_fwd = time_kernel(lambda: ops.sdf_fwd_loss_raw(feats, meta, pack, sb, step.aux, mask, step.gpred,
                                                         step.loss_slots, "L1", 1.0, 0.0, 0.0, sdf_out=None))
        t_loss = 0.0
        t_bwd = time_kernel(lambda: ops.sdf_bwd_raw(step.x, feats, meta, pack, step.gpred, mask, False,
                                                    [True] * L, step.grads, sorted_batch=sb, overwrite=True,
                                                    gsdf_sorted=True))
        if fused:
            # ... and what the step actually launches: forward + loss + decoder backward as ONE kernel, then the pull
            # (on the step's OWN binned batch -- the index rides in xn.w, no perm[] -- not on the reference one)
            t_train = time_kernel(lambda: ops.sdf_train_raw(feats, meta, pack, sb_step, step.aux, step.loss_slots,
                                                            step.grads, "L1", 1.0, 0.0, 0.0))
    else:
        t_fwd = time_kernel(lambda: ops.sdf_fwd_raw(step.x, feats, meta, pack, True, out=step.sdf, mask=mask))
        cols = [c.contiguous() for c in (step.target, step.valid, step.sign, step.weight)]
        t_loss = time_kernel(lambda: ops.mapping_loss_raw(step.sdf, *cols, "L1", 1.0, 0.0, 0.0, step.gpred, step._loss))
        t_bwd = time_kernel(lambda: ops.sdf_bwd_raw(step.x, feats, meta, pack, step.gpred, mask, False,
                                                    [True] * L, step.grads))
    t_zero = time_kernel(lambda: [g.zero_() for g in step.grads]) if sb is None else 0.0
    b_fwd = 12 + 32 * L * C + 4        # xyz + 8 corners x C x 4 B per level + sdf
    b_bwd = 4 + 32 * L * C             # dL/dsdf + grad scatter counted once as a write
    dom = ("sdf_bwd_kernel", t_bwd, b_bwd)
    t_pull = 0.0
    if sb is None:
        kernels_us = {"in_the_step": {"sdf_fwd_kernel": t_fwd, "mapping_loss_kernel": t_loss, "sdf_bwd_kernel": t_bwd,
                                      "zero_grads": t_zero}}
    else:
        # the backward is two launches: the MFMA pass that leaves the d-feat rows in the workspace,
        # and the owner-computes pull that forms the grid gradient from them; time the pull alone
        ws = sb.bwd_workspace(N_POINTS * L * C)
        t_pull = time_kernel(lambda: ops.grad_pull_raw(feats, meta, sb, ws, step.grads, overwrite=True))
        # which kernel that is (asked of the library, not assumed): the matrix-core pull, or the vector kernels
        # (MISO_PULL_MC=0, or a grid the matrix-core kernel does not take)
        from miso_amd import _lib as _l
        on_mc = bool(_l.load().miso_grad_pull_on_matrix_cores(
            ops.C.byref(ops._fill_grid(feats, meta, step.grads, data=False)), sb.tiles, N_POINTS, L * C))
        pull_name = "grad_pull_mc_kernel" if on_mc else "grad_pull_block_kernel(+drain)"
        two_launch = {"sdf_fwd_kernel(+mapping loss)": t_fwd, "sdf_bwd_kernel(MFMA pass)": t_bwd - t_pull, pull_name: t_pull}
        if t_train is not None:
            # the two-launch forward / backward are timed for reference only -- the step does not launch them
            kernels_us = {"in_the_step": {"sort_points(3 launches)": t_sort, "sdf_train_kernel": t_train - t_pull,
                                          pull_name: t_pull},
                          "not_in_the_step(two-launch form, for reference)": two_launch}
        else:
            kernels_us = {"in_the_step": {"sort_points(3 launches)": t_sort, **two_launch}}
        # algorithmic bytes of the pull: the gradient of 8 corners x C channels per level, counted
        # once as a write (SURVEY 8d backward figure without the 4 B of dL/dsdf the MFMA pass reads)
        dom = (pull_name, t_pull, 32 * L * C)
    if t_train is not None and t_train - t_pull > dom[1]:
        # the fused kernel moves the bytes of both passes except the gradient write (the pull's): corners read once
        dom = ("sdf_train_kernel", t_train - t_pull, b_fwd)
    elif t_train is None and t_fwd > dom[1]:
        dom = ("sdf_fwd_kernel", t_fwd, b_fwd)
    achieved = N_POINTS * dom[2] / (dom[1] * 1e-6) / 1e9
    # HBM traffic and MFMA busy fraction are PMC figures (rocprofv3 --pmc, separate passes: tools/profile_bench.sh ->
    # tools/pmc_summary.py -> profiles/<round>_pmc_summary.json).  They are quoted only while that file was measured
    # on the kernels of THIS tree: it carries the source hash miso_version() embeds.
    from miso_amd import _lib
    from miso_amd.csrc_hash import source_hash
    lib_ver = _lib.load().miso_version().decode()
    traffic, mfma_pmc, pmc_note = None, None, "no PMC summary for these kernel sources (run tools/profile_bench.sh)"
    import glob
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            js = json.load(open(pmc))
        except Exception:
            continue
        if js.get("_meta", {}).get("source_hash") == source_hash():
            if dom[0].startswith("grad_pull"):          # kernel_us covers every launch of the pull: so does the traffic
                parts = [v.get("hbm_bytes_per_launch") for k, v in js.items() if k.startswith("grad_pull") and isinstance(v, dict)]
                traffic = sum(p for p in parts if p is not None) if parts else None
            else:
                traffic = js.get(dom[0], {}).get("hbm_bytes_per_launch")
            mfma_pmc = {k: v["mfma_busy_frac"] for k, v in js.items() if isinstance(v, dict) and "mfma_busy_frac" in v}
            pmc_note = f"{os.path.basename(pmc)} (kernel sources {source_hash()}, commit {js['_meta'].get('commit')})"
            break
    # fp32 MFMA share of the two fused kernels, from the live kernel times: the decoder's matrix FLOPs (frozen
    # decoder: forward 2(F H + H H + H), backward the same w.r.t. activations) over v_mfma_f32_32x32x2_f32 peak
    F_ = L * C
    mlp_flop = 2.0 * (F_ * HIDDEN + HIDDEN * HIDDEN + HIDDEN) * N_POINTS
    t_bwd_mfma = t_bwd - t_pull
    exact = bool(ops._EXACT_F32) or os.environ.get("MISO_EXACT_F32", "0") not in ("", "0")
    if exact:
        # exact fp32 chains (v_mfma_f32_32x32x2_f32): the decoder's FLOPs over the fp32 matrix peak
        peak, flop_f, flop_b = 157.3, mlp_flop, mlp_flop
        definition = "decoder matrix FLOPs / kernel time / fp32 MFMA peak (live, HIP events)"
    else:
        # bf16x3 split products (round 6): what the matrix cores are ISSUED -- six 32x32x16 piece products per k-block of 16
        # in the forward layers and the last backward product, three in the first backward one (its B operand is a 0 / 1
        # mask) -- over the dense bf16 peak.  (The decoder's own FLOPs over the fp32 matrix peak: `nominal_fp32`.)
        RT, KB0, KBH = HIDDEN // 32, (F_ + 15) // 16, HIDDEN // 16
        per_mfma = 32 * 32 * 16 * 2.0
        peak = 2500.0
        flop_f = RT * 6 * (KB0 + KBH) * per_mfma / 32 * N_POINTS
        flop_b = (RT * 3 * KBH + 6 * KBH) * per_mfma / 32 * N_POINTS
        definition = ("bf16 piece-product FLOPs issued to the matrix cores / kernel time / dense bf16 MFMA peak (live, HIP "
                      "events); nominal_fp32 = the decoder's own FLOPs / kernel time / fp32 MFMA peak (157.3)")
    mfma = {"peak_TFLOPs": peak, "arithmetic": "exact fp32 chains" if exact else "bf16x3 split products, fp32 accumulate",
            "sdf_fwd_kernel": flop_f / (t_fwd * 1e-6) / (peak * 1e12),
            "sdf_bwd_kernel": flop_b / (t_bwd_mfma * 1e-6) / (peak * 1e12),
            **({"sdf_train_kernel": (flop_f + flop_b) / ((t_train - t_pull) * 1e-6) / (peak * 1e12)} if t_train is not None else {}),
            "definition": definition}
    if not exact:
        mfma["nominal_fp32"] = {"sdf_fwd_kernel": mlp_flop / (t_fwd * 1e-6) / 157.3e12,
                                "sdf_bwd_kernel": mlp_flop / (t_bwd_mfma * 1e-6) / 157.3e12,
                                **({"sdf_train_kernel": 2 * mlp_flop / ((t_train - t_pull) * 1e-6) / 157.3e12}
                                   if t_train is not None else {})}
    if mfma_pmc:
        mfma["pmc_busy_frac"] = mfma_pmc
    # Both roofs of the dominant kernel, and which one binds: the fused kernels run the decoder on v_mfma_f32_32x32x2_f32
    # (exact fp32, 157.3 TFLOP/s), and at cfg-2 that roof is closer than HBM's (0.52 vs 0.35)
    frac_hbm = achieved / 8000.0
    frac_mfma = mfma.get(dom[0])
    if frac_mfma is not None and frac_mfma > frac_hbm:
        fpp = {"sdf_fwd_kernel": flop_f, "sdf_bwd_kernel": flop_b, "sdf_train_kernel": flop_f + flop_b}[dom[0]] / N_POINTS
        roofline = {"bound": "mfma", "kernel": dom[0], "achieved": frac_mfma * peak, "peak": peak, "unit": "TFLOP/s",
                    "frac": frac_mfma, "flop_per_point": fpp}
    else:
        roofline = {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": frac_hbm}
    roofline.update({"frac_hbm": frac_hbm, "frac_mfma": frac_mfma, "achieved_hbm_GBps": achieved, "peak_hbm_GBps": 8000.0,
                     "traffic": traffic, "traffic_source": pmc_note, "algorithmic_bytes_per_point": dom[2],
                     "kernel_us": dom[1], "mfma_frac": mfma})
    if dom[0].startswith("grad_pull"):
        roofline["launches"] = ("grad_pull_mc_kernel: one launch (crowded blocks are worked off in epochs inside it)"
                                if dom[0] == "grad_pull_mc_kernel" else
                                "grad_pull_block_kernel + its drain launch for queued slices of crowded tiles")

    out = {
        "metric": "3D point-samples/sec (encode+decode fwd+bwd), 262144-pt batch",
        "value": value, "unit": "point-samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "settle_steps": SETTLE_STEPS, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (exact fp32 decoder chains)" if exact else
                 "f32 (decoder products as bf16x3 pieces on the 16-bit matrix cores, fp32 accumulate: error against float64 "
                 "equal to the exact fp32 chains', tests/test_split_precision.py; MISO_F_EXACT_F32 selects those)",
        "data": "synthetic",
        "config": {"workload": "cfg-2: one submap per GPU, 3-level {32,64,128}^3 grid C=8 + MLP 24-64-64-1 "
                               "(frozen, seeded random weights), 262144 uniform-in-bbox points per GPU per "
                               "step, L1 loss, grads to all levels",
                   "points_per_gpu": N_POINTS, "levels": list(LEVELS), "feature_dim": C,
                   "decoder": [C * L, HIDDEN, HIDDEN, 1], "parallelism": f"submap-parallel x{world}",
                   "launch": "one graph replay per step" if step._use_graph else
                             "five stream launches per step (a graph replay leaves the device idle ~6 us between replays)"},
        "roofline": roofline,
        "step_fraction_of_hbm_roofline": value / world * (20 + 64 * L * C) / 8e12,
        "kernels_us": kernels_us,
        "binned": sb is not None,
        "library": lib_ver,
        "step_note": "a training step needs the loss and the gradients, not the per-point SDF in caller order "
                     "(keep_sdf=False): the 4 B/point SDF write of SURVEY 8(d)'s B_alg is not moved by this step "
                     "(1 MB of 408 MB); step_fraction_of_hbm_roofline uses the full B_alg = 20 + 64 L C",
    }
    return out


if __name__ == "__main__":
    main()
