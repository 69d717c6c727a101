"""Dev: wall time per alignment iteration, captured loop vs op-by-op (same setup as tools/align_bench.py)."""
import logging
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.basicConfig(level=logging.WARNING)
import miso_amd.grid_opt.align.base as AB  # noqa: E402
import miso_amd.grid_opt.align.miso as AM  # noqa: E402
from miso_amd.grid_opt.models.grid_atlas import GridAtlas  # noqa: E402

dev = torch.device("cuda", 0)
cfg = {"name": "grid_net", "spatial_dim": 3,
       "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True, "fix": True,
                   "pretrained_model": None},
       "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 1e-2, "bound": [[-10., 10.], [-5., 5.], [-10., 10.]],
                "base_cell_size": 0.5, "per_level_scale": 5, "n_levels": 2},
       "pose": {"optimize": False, "num_poses": 1}}
torch.manual_seed(0)
atlas = GridAtlas(cfg, device=dev)
lb = torch.tensor(cfg["grid"]["bound"])
for s_, tx in enumerate((0.0, 9.0, 4.0)):
    atlas.add_submap(lb, torch.eye(3), torch.tensor([[tx], [0.3], [-0.4]]), num_poses=1)
    atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
atlas.to(dev)
atlas.precompute_coordinates_for_alignment()


class DS(torch.utils.data.Dataset):
    def __len__(self):
        return 1

    def __getitem__(self, i):
        return 0


def latent(at, ld, a, b):
    return AM.pairwise_loss_latent(at, ld, a, b, level=1, fdim=4, align_loss="L2", device=dev)


latent.device_gate = True
latent.batched = lambda at, pairs, chk: AM.pairwise_loss_latent_batched(at, pairs, level=1, fdim=4, check_intersection=chk,
                                                                        device=dev)
for captured in (True, False):
    atlas.no_captured_alignment = not captured
    AB.generic_align_multiple_submaps(atlas, DS(), ("w", latent), num_iters=9, lr=1e-3, verbose=False)
    res = {}
    for iters in (20, 120):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        AB.generic_align_multiple_submaps(atlas, DS(), ("t", latent), num_iters=iters - 1, lr=1e-3, verbose=False)
        torch.cuda.synchronize()
        res[iters] = time.perf_counter() - t0
    per = (res[120] - res[20]) / 100 * 1e6
    print(f"captured={captured}: {res[20] * 1e3:.1f} ms for 20 iterations, {res[120] * 1e3:.1f} ms for 120 -> "
          f"{per:.0f} us per extra iteration (3 pairs, 4 M vertices each)")
