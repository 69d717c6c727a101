"""dev: cProfile of the host side of GridTrainer.train_step at the Newer College shape (cfg-5: 6 144 samples, the step
is host-bound).  GRID / N as tools/trainer_bench.py."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miso_amd.grid_opt.loss as L  # noqa: E402
from miso_amd.grid_opt.models.grid_net import GridNet  # noqa: E402
from miso_amd.grid_opt.trainer import GridTrainer  # noqa: E402

dev = "cuda:0"
n = int(os.environ.get("N", 6144))
cfg = {"name": "grid_net", "spatial_dim": 3,
       "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                   "fix": True, "pretrained_model": None},
       "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 1e-2, "bound": [[-60., 60.], [-60., 60.], [-5., 15.]],
                "base_cell_size": 1.0, "per_level_scale": 5, "n_levels": 2},
       "pose": {"optimize": False, "num_poses": 1}}
g = torch.Generator().manual_seed(1)
x = (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([25.0, 25.0, 4.0]) + torch.tensor([5.0, -8.0, 2.0])
torch.manual_seed(0)
batch = ({"coords_frame": x[None].to(dev), "sample_frame_ids": torch.zeros(1, n, 1, dtype=torch.int64, device=dev),
          "weights": torch.ones(1, n, 1, device=dev)},
         {"sdf": (torch.rand(1, n, 1, generator=g) * 0.2 - 0.1).to(dev), "sdf_valid": torch.ones(1, n, 1, device=dev),
          "sdf_signs": torch.zeros(1, n, 1, device=dev)})
net = GridNet(cfg, device=dev).to(dev)
net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
net.unlock_feature()
net.lock_pose()
tcfg = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1, "ckpt_every": -1,
        "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": tempfile.mkdtemp(),
        "relchange_tol": 0, "max_epochs_in_level": 1000, "grid_training_mode": "joint"}
lossf = L.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
tr = GridTrainer(tcfg, net, lossf, None, None, dev, torch.float32)
for _ in range(200):
    tr.train_step(*batch)
torch.cuda.synchronize()
import gc
gc.disable()
t0 = time.perf_counter()
for _ in range(200):
    tr.train_step(*batch)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host {t_host / 200 * 1e6:.1f} us per step, wall {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    tr.train_step(*batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
