"""dev: where the wall time of one sharded alignment run goes (2 ranks on one GPU over gloo, or 1 rank over nccl).
   MISO_BENCH_BACKEND=gloo python tools/dist_probe.py 2      (spawns the ranks itself)"""
import cProfile
import os
import pstats
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    if "RANK" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(world)],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                                           MASTER_PORT=str(port))) for r in range(world)]
        raise SystemExit(max(p.wait() for p in procs))
    import torch
    import torch.distributed as dist
    import bench
    from miso_amd import dist as mdist
    import miso_amd.grid_opt.align.miso as AM
    rank = int(os.environ["RANK"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    backend = os.environ.get("MISO_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    atlas = bench.scannet_atlas(dev, 8)
    atlas.precompute_coordinates_for_alignment()

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return 0

    for level in (0, 1):
        loss = AM.latent_loss_for_level(atlas, level, device=dev)

        def run(n):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            mdist.align_multiple_submaps_distributed(atlas, DS(), (f"latent{level}", loss), num_iters=n - 1, lr=0.0,
                                                     verbose=True, save_iterations=True, always_reduce=True)
            torch.cuda.synchronize()
            dist.barrier()
            return time.perf_counter() - t0
        run(4)
        ts = [run(20) for _ in range(3)] + [run(120) for _ in range(2)]
        if rank == 0:
            print(f"level {level}: run(20) {[round(t * 1e3, 1) for t in ts[:3]]} ms, run(120) {[round(t * 1e3, 1) for t in ts[3:]]} ms")
            pr = cProfile.Profile()
            pr.enable()
        run(20)
        if rank == 0:
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
