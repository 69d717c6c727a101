"""Dev tool: one ScanNet-shaped submap (cfg-3) trained straight from depth frames -- wall time per iteration of
GridTrainer with the exact-size dataset (row count read back, step re-captured when it changes) and with the
padded one (fixed capacity, one captured step)."""
import sys, time, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import golden_cases as gc
from miso_amd.grid_opt.datasets.sdf_rgbd import PosedSdfRgbd
from miso_amd.grid_opt.loss import MisoLossMapping
from miso_amd.grid_opt.models.grid_net import GridNet
from miso_amd.grid_opt.trainer import GridTrainer
from miso_amd.grid_opt.utils.utils_data import CameraParameters
from miso_amd.grid_opt.utils.utils import collate_batch_of_one
dev = 'cuda:0'
B, H, W, rays = 100, 480, 640, 200
g = torch.Generator().manual_seed(3)
depth = torch.rand(B, H, W, generator=g) * 3.0 + 0.5
depth[torch.rand(B, H, W, generator=g) < 0.1] = 0.0
ang = torch.rand(B, generator=g) * 6.28
R = torch.eye(3).repeat(B, 1, 1)
R[:, 0, 0], R[:, 0, 2], R[:, 2, 0], R[:, 2, 2] = ang.cos(), ang.sin(), -ang.sin(), ang.cos()
t = (torch.rand(B, 3, 1, generator=g) - 0.5) * torch.tensor([[10.0], [4.0], [10.0]])
cp = CameraParameters(fx=577.6, fy=578.7, cx=318.9, cy=242.7, H=H, W=W)
cfg_model = gc.model_cfg([[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]], 0.5, 5, 2, 4, 64, num_poses=B, init_stddev=1e-2)
lf = MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=1.0, trunc_dist=0.15)
import os
for padded in ((True,) if os.environ.get('MISO_E2E_ONLY_PADDED') else (False, True)):
    ds = PosedSdfRgbd.from_frames(depth, R, t, cp, n_rays=rays, n_strat_samples=19, n_surf_samples=8, trunc_dist=0.15,
                                  device=dev, normals=torch.ones(B, H, W, 3), padded=padded)
    torch.manual_seed(0)
    net = GridNet(cfg_model, device=dev).to(dev)
    for k in range(B):
        net.set_initial_kf_pose(k, R[k], t[k], kf_key=f"KF{k}")
    net.unlock_feature()
    net.lock_pose()
    cfg_train = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-2, "epochs": 1, "ckpt_every": -1,
                 "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": "/tmp/e2e",
                 "relchange_tol": 0, "max_epochs_in_level": 1000, "grid_training_mode": "joint"}
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0, collate_fn=collate_batch_of_one)
    tr = GridTrainer(cfg_train, net, lf, loader, None, dev, torch.float32)
    tr.total_steps, tr.total_epoch_time = 0, 0
    for e in range(5):
        tr.train_epoch(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_it = 50
    for e in range(n_it):
        tr.train_epoch(e)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_it
    rows = B * rays * 27
    print(f"padded={padded}: {dt * 1e3:.3f} ms / iteration ({rows} row capacity, {rows / dt / 1e6:.1f} M rows/s)")

# ---- where the padded iteration's time goes (each piece synchronised) ---------------------------------------
from miso_amd.grid_opt.utils.utils import prepare_batch


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print(f"  {name:34s} {(time.perf_counter() - t0) / n * 1e6:9.1f} us")
    return r


timeit("dataset[0]", lambda: ds[0])
it = iter(loader)
timeit("DataLoader next (collate)", lambda: next(iter(loader)))
mi, gt = next(iter(loader))
mi, gt = timeit("prepare_batch", lambda: prepare_batch(mi, gt, dev))
step = next(iter(tr._mapping_steps.values()))
with torch.no_grad():
    xw = timeit("world_coords", lambda: lf.world_coords(net, mi['coords_frame'][0], mi['sample_frame_ids'][0, :, 0]))
    timeit("set_batch", lambda: step.set_batch(xw, gt['sdf'][0], gt['sdf_valid'][0], gt['sdf_signs'][0],
                                               mi['weights'][0], live_rows=mi['live_rows']))
timeit("step.run (graph replay)", step.run)
timeit("optimizer.step", tr.optimizer.step)
timeit("train_step", lambda: tr.train_step(mi, gt))
