"""Wall time of Tracker.lm_step / track_window at a SLAM-sized batch (dev)."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd.grid_opt.models.grid_net import GridNet  # noqa: E402
from miso_amd.grid_opt.slam.tracker import Tracker  # noqa: E402

dev = "cuda:0"
n = int(os.environ.get("N", 16384))
cfg_m = {"name": "grid_net", "spatial_dim": 3,
         "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                     "fix": True, "pretrained_model": None},
         "grid": {"type": "regular", "feature_dim": 4, "init_stddev": 1e-2, "bound": [[-10., 10.], [-5., 5.], [-10., 10.]],
                  "base_cell_size": 0.5, "per_level_scale": 5, "n_levels": 2},
         "pose": {"optimize": True, "num_poses": 2}}
torch.manual_seed(0)
net = GridNet(cfg_m, device=dev).to(dev)
net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
net.set_initial_kf_pose(1, torch.eye(3), torch.tensor([[0.1], [0.0], [0.05]]), kf_key="KF1")
g = torch.Generator().manual_seed(1)
pts = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([18.0, 9.0, 18.0])
sdf = torch.rand(n, 1, generator=g) * 0.2 - 0.1


class DS(torch.utils.data.Dataset):
    def select_keyframes(self, kfs):
        pass

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return ({"coords_frame": pts, "sample_frame_ids": torch.ones(n, 1, dtype=torch.int64),
                 "weights": torch.ones(n, 1)},
                {"sdf": sdf, "sdf_valid": torch.ones(n, 1), "sdf_signs": torch.zeros(n, 1)})


cfg = {"device": dev, "train": {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1,
                                "ckpt_every": -1, "eval_every": -1, "eval_metric": None, "pretrained_model": None,
                                "log_dir": tempfile.mkdtemp()},
       "tracking": {"learning_rate": 1e-3, "verbose": False, "gm_scale_sdf": 0.1, "lm_lambda": 5.0, "lm_max_iter": 3,
                    "lm_tol_deg": 0.0, "lm_tol_m": 0.0, "loss_type": "GM", "trunc_dist": 0.3, "solver": "lm"}}
trk = Tracker(net, DS(), cfg)
for _ in range(3):
    trk.lm_step(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    trk.lm_step(1)
torch.cuda.synchronize()
print(f"lm_step N={n}: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us")
cfg["tracking"]["solver"] = "adam"
trk = Tracker(net, DS(), cfg)
trk.track_window([1], iterations=3)
torch.cuda.synchronize()
t0 = time.perf_counter()
trk.track_window([1], iterations=20)
torch.cuda.synchronize()
print(f"track_window (Adam) N={n}: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per iteration")
if os.environ.get("PROFILE"):
    import cProfile
    import pstats
    cfg["tracking"]["solver"] = "lm"
    trk = Tracker(net, DS(), cfg)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        trk.lm_step(1)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    cfg["tracking"]["solver"] = "adam"
    trk = Tracker(net, DS(), cfg)
    pr = cProfile.Profile()
    pr.enable()
    trk.track_window([1], iterations=10)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
