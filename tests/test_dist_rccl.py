"""The multi-rank path on the single-GPU box.

(a) RCCL itself: a process group of ONE rank with backend "nccl" (= RCCL on ROCm) bound to the device, as bench.py and
    the driver's 8-GPU run create it -- `init_process_group("nccl", device_id=...)`, an all-reduce and a broadcast of
    device memory, the fused alignment loop with the all-reduce hook LIVE between its two captured halves (graph A ->
    all_reduce on the same stream -> graph B), against the captured single-process loop.  What a second GPU adds is
    peers, not code paths.
(b) bench.py's watchdog: two ranks (gloo, sharing the device: MISO_BENCH_BACKEND / MISO_BENCH_DEVICE), rank 1 made to
    throw inside the cfg-3 extra -- the headline line must still come out, with the error recorded, well inside
    three minutes instead of after a collective timeout.
Both run in child processes (a process group and an os._exit are process-wide)."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RCCL_SCRIPT = r"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
from miso_amd import dist as mdist
from test_grid_opt_mirror import make_atlas
import miso_amd.grid_opt.align.base as AB
import miso_amd.grid_opt.align.miso as AM

# all-reduce / broadcast of device memory through RCCL
t = torch.arange(19, device=dev, dtype=torch.float32)
mdist.all_reduce_sum(t, always=True)
torch.cuda.synchronize()
assert torch.equal(t.cpu(), torch.arange(19, dtype=torch.float32))
atlas = make_atlas("cuda:0")
before = [p.detach().clone() for s in range(atlas.num_submaps) for p in atlas.get_submap(s).parameters()]
mdist.sync_submaps(atlas, always=True)
torch.cuda.synchronize()
after = [p.detach() for s in range(atlas.num_submaps) for p in atlas.get_submap(s).parameters()]
assert all(torch.equal(a, b) for a, b in zip(before, after))
assert all(g.feature.is_contiguous(memory_format=torch.channels_last_3d) for s in range(atlas.num_submaps)
           for g in atlas.get_submap(s).features)

class DS(torch.utils.data.Dataset):
    def __len__(self): return 1
    def __getitem__(self, i): return 0

def poses(a):
    return torch.cat([torch.cat((r.detach().reshape(1, 3), t.detach().reshape(1, 3)), 1)
                      for r, t in zip(a.rotation_corrections, a.translation_corrections)]).cpu()

out = {}
for mode in ("single", "rccl"):
    a = make_atlas("cuda:0")
    a.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    snaps = []
    for l in range(2):
        tup = (f"latent{l}", AM.latent_loss_for_level(a, l, align_loss="L2", device="cuda:0"))
        kw = dict(num_iters=12, lr=1e-2, pose_reg_weight=1.0, pose_thresh_rad=1e-3, pose_thresh_m=1e-3,
                  save_iterations=True)
        if mode == "single":
            info = AB.generic_align_multiple_submaps(a, DS(), tup, verbose=False, **kw)
        else:
            info = mdist.align_multiple_submaps_distributed(a, DS(), tup, always_reduce=True, **kw)
        snaps.append(torch.stack([info["iteration_results"][i] for i in range(13)]).cpu())
    out[mode] = (poses(a), snaps)
torch.cuda.synchronize()
d = (out["single"][0] - out["rccl"][0]).abs().max().item()
ds = max((x - y).abs().max().item() for x, y in zip(out["single"][1], out["rccl"][1]))
print("RCCL_OK", d, ds)
assert d <= 1e-6 and ds <= 1e-6, (d, ds)
dist.destroy_process_group()
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_rccl_single_rank_group_runs_the_collectives_and_the_sharded_loop(tmp_path):
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(RCCL_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, str(script), ROOT, str(_free_port())], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
def test_bench_headline_survives_a_rank_that_fails_inside_an_extra():
    env = dict(os.environ, MISO_BENCH_BACKEND="gloo", MISO_BENCH_DEVICE="0", MISO_BENCH_FAIL="1:map_cfg3")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3"],
                         capture_output=True, text=True, timeout=400, env=env)
    took = time.time() - t0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["roofline"]["frac"] > 0
    assert "injected failure in map_cfg3" in rec["extras_multi_gpu"]["error"]
    assert took < 180, took
