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



# ---- the extras live in tools/bench_extras.py; re-exported here (tests and tools reach them through `bench`) -----------
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_extras import (SCANNET_CFG, align_cfg4, atlas_mesh_extraction, cfg4_pmc_traffic, eikonal_step, extras,  # noqa: E402,F401
                          extras_multi, map_cfg3, mesh_extraction, sample_generation, scannet_atlas, slam_components,
                          trainer_steps)


if __name__ == "__main__":
    main()
