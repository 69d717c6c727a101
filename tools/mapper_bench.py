"""Wall time of Mapper.mapping (a fresh GridTrainer per call, as the SLAM loop does per frame) at the ScanNet shape (dev)."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd.grid_opt.models.grid_net import GridNet  # noqa: E402
from miso_amd.grid_opt.slam.mapper import Mapper  # noqa: E402

dev = "cuda:0"
n = int(os.environ.get("N", 540000))
iters = int(os.environ.get("ITERS", 10))
mode = os.environ.get("MODE", "coordinate+joint")
cfg_m = {"name": "grid_net", "spatial_dim": 3,
         "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                     "fix": True, "pretrained_model": None},
         "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 1e-2, "bound": [[-10., 10.], [-5., 5.], [-10., 10.]],
                  "base_cell_size": 0.5, "per_level_scale": 5, "n_levels": 2},
         "pose": {"optimize": True, "num_poses": 4}}
torch.manual_seed(0)
net = GridNet(cfg_m, device=dev).to(dev)
for k in range(4):
    net.set_initial_kf_pose(k, torch.eye(3), torch.zeros(3, 1), kf_key=f"KF{k}")
g = torch.Generator().manual_seed(1)
pts = ((torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([6.0, 2.5, 6.0])).to(dev)
sdf = (torch.rand(n, 1, generator=g) * 0.2 - 0.1).to(dev)
ids = torch.randint(0, 4, (n, 1), generator=g).to(dev)
one, zero = torch.ones(n, 1, device=dev), torch.zeros(n, 1, device=dev)


class DS(torch.utils.data.Dataset):
    def select_keyframes(self, kfs):
        pass

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return ({"coords_frame": pts, "sample_frame_ids": ids, "weights": one},
                {"sdf": sdf, "sdf_valid": one, "sdf_signs": zero})


cfg = {"device": dev,
       "train": {"trainer": "base", "verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 50,
                 "batch_size": 1000000, "ckpt_every": -1, "eval_every": -1, "pretrained_model": None,
                 "log_dir": tempfile.mkdtemp(), "relchange_tol": 0, "max_epochs_in_level": 100,
                 "grid_training_mode": mode},
       "mapping": {"learning_rate": 1e-3, "loss_type": "L1", "weight_sdf": 1.0, "weight_eik": 0.0, "weight_fs": 0.1,
                   "trunc_dist": 0.15, "finite_diff_eps": 0.01, "grad_method": "finitediff", "eik_trunc_dist": 0.024,
                   "verbose": False}}
mp = Mapper(net, DS(), cfg)
for _ in range(2):
    mp.mapping([0, 1, 2, 3], iterations=iters, level_iterations=5)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    mp.mapping([0, 1, 2, 3], iterations=iters, level_iterations=5)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"Mapper.mapping N={n} iterations={iters} mode={mode}: {dt * 1e3:.2f} ms per call = {dt / iters * 1e6:.0f} us per iteration")
