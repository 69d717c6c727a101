"""Wall time per GridTrainer.train_step at cfg-2 (dev): captured graph replay vs op-by-op autograd."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miso_amd.grid_opt.loss as L  # noqa: E402
from miso_amd.grid_opt.models.grid_net import GridNet  # noqa: E402
from miso_amd.grid_opt.trainer import GridTrainer  # noqa: E402

dev = "cuda:0"
if os.environ.get("STREAM_MIN") is not None:      # dev: force stream launches (0) / graph replays (huge) whatever n
    from miso_amd.step import MappingStep
    MappingStep.STREAM_MIN_POINTS = int(os.environ["STREAM_MIN"])
n = int(os.environ.get("N", 262144))
cfg = {"name": "grid_net", "spatial_dim": 3,
       "decoder": {"type": "mlp", "hidden_dim": 64, "hidden_layers": 1, "out_dim": 1, "pos_invariant": True,
                   "fix": True, "pretrained_model": None},
       "grid": {"type": "regular", "feature_dim": 8, "init_stddev": 1e-2, "bound": [[-1., 1.]] * 3,
                "base_cell_size": 2.0 / 32, "per_level_scale": 2, "n_levels": 3},
       "pose": {"optimize": False, "num_poses": 1}}
shape = os.environ.get("GRID", "cfg2")
g = torch.Generator().manual_seed(1)
x = torch.rand(n, 3, generator=g) * 2 - 1
if shape == "ncd":        # cfg-5, Newer College quad: 120 x 120 x 20 m, cells 1.0 / 0.2 m, C=4 (configs/lidar/ncd_quad.yaml)
    cfg["grid"].update(feature_dim=4, bound=[[-60., 60.], [-60., 60.], [-5., 15.]], base_cell_size=1.0, per_level_scale=5,
                       n_levels=2)
    x = x * torch.tensor([25.0, 25.0, 4.0]) + torch.tensor([5.0, -8.0, 2.0])     # the sensor's surroundings, not the whole bound
elif shape == "scannet":  # cfg-3: 20 x 10 x 20 m, cells 0.5 / 0.1 m, C=4
    cfg["grid"].update(feature_dim=4, bound=[[-10., 10.], [-5., 5.], [-10., 10.]], base_cell_size=0.5, per_level_scale=5,
                       n_levels=2)
    x = x * torch.tensor([6.0, 2.5, 6.0])
torch.manual_seed(0)
batch = ({"coords_frame": x[None].to(dev), "sample_frame_ids": torch.zeros(1, n, 1, dtype=torch.int64, device=dev),
          "weights": torch.ones(1, n, 1, device=dev)},
         {"sdf": (torch.rand(1, n, 1, generator=g) * 0.2 - 0.1).to(dev), "sdf_valid": torch.ones(1, n, 1, device=dev),
          "sdf_signs": torch.zeros(1, n, 1, device=dev)})
for captured in (True, False):
    net = GridNet(cfg, device=dev).to(dev)
    net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
    net.unlock_feature()
    net.lock_pose()
    tcfg = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1, "ckpt_every": -1,
            "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": tempfile.mkdtemp(),
            "relchange_tol": 0, "max_epochs_in_level": 1000, "grid_training_mode": "joint", "captured_step": captured}
    lossf = L.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
    tr = GridTrainer(tcfg, net, lossf, None, None, dev, torch.float32)
    for _ in range(5):
        tr.train_step(*batch)
        if os.environ.get("SHOW_ACTIVE"):
            torch.cuda.synchronize()
            print("   step", _, [(float((p_.grad != 0).float().mean()), int(torch.isnan(p_.grad).sum()), float(p_.grad.abs().max()))
                                 for p_ in tr.optimizer.state if p_.grad is not None and p_.dim() == 5])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 50
    for _ in range(iters):
        tr.train_step(*batch)
        if os.environ.get("SHOW_ACTIVE") == "2" and _ % 5 == 0:
            torch.cuda.synchronize()
            print("   step", _, [(float((p_.grad != 0).float().mean()), int(torch.isnan(p_.grad).sum()), float(p_.grad.abs().max()))
                                 for p_ in tr.optimizer.state if p_.grad is not None and p_.dim() == 5])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"   host side of the loop: {t_host / iters * 1e6:.0f} us per step")
    if os.environ.get("SHOW_ACTIVE"):
        for prm, st in tr.optimizer.state.items():
            if "active" in st:
                print("  param", tuple(prm.shape), "active chunks", float(st["active"].float().mean()),
                      "nonzero exp_avg", float((st["exp_avg"] != 0).float().mean()),
                      "nonzero grad", float((prm.grad != 0).float().mean()) if prm.grad is not None else None)
    print(f"train_step N={n} captured={captured}: {(time.perf_counter() - t0) / iters * 1e6:.0f} us per step "
          f"({n / ((time.perf_counter() - t0) / iters) / 1e6:.0f} Mpts/s incl. Adam over all levels)")
