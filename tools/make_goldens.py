"""Generate tests/golden/*.npz by IMPORTING the reference (MISO) on CPU.

Runs only in the build container (needs /root/reference).  The reference can
not travel to the GPU box, so its outputs on seed-pinned inputs
(tests/golden_cases.py) are committed as small fixtures instead.

Import recipe: stub packages for GUI / IO dependencies that are
absent from the image and irrelevant to the arithmetic (open3d, trimesh, ...);
pytorch3d.transforms is provided by tools/ref_shims (so3_exp_map / hat restated
-> that boundary is "parity unpinned").  Harness-only patches: PerfTimer (needs
torch.cuda.Event) and prepare_batch's default device.

    python tools/make_goldens.py            # writes tests/golden/*.npz
"""
import importlib.abc
import importlib.machinery
import logging
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MISO_REFERENCE", "/root/reference")

STUB_ROOTS = {"open3d", "trimesh", "mcubes", "cv2", "torchvision", "evo", "pysdf", "sdf",
              "lpips", "dearpygui"}
EXTRA = {"pytorch3d.ops", "torch.utils.tensorboard"}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS or fullname in EXTRA:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, os.path.join(HERE, "ref_shims"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    logging.disable(logging.CRITICAL)
    import grid_opt.utils.utils as ru

    class _Timer:  # harness patch: upstream PerfTimer needs torch.cuda.Event
        def __init__(self, *a, **k):
            pass

        def reset(self):
            pass

        def check(self, *a, **k):
            return 0.0, 0.0

    ru.PerfTimer = _Timer
    ru.prepare_batch.__defaults__ = ("cpu",)
    ru.get_batch.__defaults__ = ("cpu",)
    return ru


import numpy as np  # noqa: E402
import torch  # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build_gridnet(GridNet, gc, case, num_poses=1, optimize_pose=False, stability=False):
    cfg = gc.model_cfg(case["bound"], case["base_cell"], case["scale"], case["n_levels"],
                       case["fdim"], case["hidden"], num_poses=num_poses,
                       optimize_pose=optimize_pose)
    net = GridNet(cfg, device="cpu")
    feats = gc.make_features(case)
    with torch.no_grad():
        for l, f in enumerate(feats):
            assert tuple(net.features[l].feature.shape) == f.shape, (net.features[l].feature.shape, f.shape)
            net.features[l].feature.copy_(T(f))
        if stability:
            for l, f in enumerate(gc.make_stability(case)):
                net.feature_stability[l].feature.copy_(T(f))
    sd = {k: T(v) for k, v in gc.make_decoder(case).items()}
    net.decoder.load_state_dict(sd)
    return net


def gen_encode_decode(name, GridNet, rloss, gc):
    case = gc.CASES[name]
    net = build_gridnet(GridNet, gc, case, stability=True)
    net.unlock_feature()
    x = T(gc.make_points(case)).requires_grad_(True)
    n = x.shape[0]
    sdf_t, valid, sign, weight = [T(a) for a in gc.make_targets(case, n)]
    feats = net.query_feature(x)
    stab = net.query_stability(x)
    pred = net(x)
    l1 = rloss.miso_loss_regression(pred, sdf_t, valid, weight, "L1")
    fs = rloss.miso_loss_free_space(pred, sdf_t, sign, 0.15)
    loss = l1 + 0.1 * fs
    params = [g.feature for g in net.features]
    grads = torch.autograd.grad(loss, params + [x])
    # second scalar with a smooth (L2) loss: better conditioned gradient check
    pred2 = net(x)
    l2 = rloss.miso_loss_regression(pred2, sdf_t, valid, weight, "L2")
    grads2 = torch.autograd.grad(l2, params + [x])
    out = dict(feats=feats.detach().numpy(), stab=stab.detach().numpy(),
               sdf=pred.detach().numpy(), loss_l1=l1.item(), loss_fs=fs.item(),
               loss_l2=l2.item(), grad_x=grads[-1].numpy(), grad2_x=grads2[-1].numpy())
    for l in range(case["n_levels"]):
        g = grads[l].numpy().reshape(-1)
        g2 = grads2[l].numpy().reshape(-1)
        idx = gc.sample_indices(g.size, 512, seed=7 + l)
        # nonzero entries are sparse in big grids: also store the top-|g| ones
        top = np.argsort(-np.abs(g))[:256].astype(np.int64)
        out[f"gfeat{l}_sum"] = np.float64(g.astype(np.float64).sum())
        out[f"gfeat{l}_abssum"] = np.float64(np.abs(g.astype(np.float64)).sum())
        out[f"gfeat{l}_idx"] = np.concatenate([idx, top])
        out[f"gfeat{l}_val"] = g[out[f"gfeat{l}_idx"]]
        out[f"g2feat{l}_abssum"] = np.float64(np.abs(g2.astype(np.float64)).sum())
        out[f"g2feat{l}_val"] = g2[out[f"gfeat{l}_idx"]]
        if name == "small":
            out[f"gfeat{l}_full"] = grads[l].numpy()
            out[f"g2feat{l}_full"] = grads2[l].numpy()
    np.savez_compressed(gc.golden_path(name), **out)
    print(f"[{name}] N={n} loss_l1={l1.item():.6f} fs={fs.item():.6f} l2={l2.item():.6e}")


def build_atlas(GridAtlas, gc):
    c = gc.ATLAS
    cfg = gc.model_cfg(c["bound"], c["base_cell"], c["scale"], c["n_levels"], c["fdim"],
                       c["hidden"])
    atlas = GridAtlas(cfg, device="cpu")
    subs = gc.atlas_inputs()
    dec = {k: T(v) for k, v in gc.make_decoder(c).items()}
    for s, sub in enumerate(subs):
        atlas.add_submap(torch.tensor(c["bound"], dtype=torch.float32), T(sub["R"]), T(sub["t"]),
                         num_poses=2)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
        net = atlas.get_submap(s)
        with torch.no_grad():
            for l, f in enumerate(sub["features"]):
                net.features[l].feature.copy_(T(f))
        net.decoder.load_state_dict(dec)
        atlas.set_submap_pose_correction(s, T(sub["dr"]), T(sub["dt"]))
    return atlas


def gen_atlas(GridAtlas, miso, rbase, gc):
    atlas = build_atlas(GridAtlas, gc)
    c = gc.ATLAS
    out = {}
    xw = T(gc.atlas_world_points())
    out["forward"] = atlas(xw).detach().numpy()
    out["query_feature"] = atlas.query_feature(xw).detach().numpy()
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    for s in range(c["n_submaps"]):
        for l in range(c["n_levels"]):
            out[f"ncoords_s{s}_l{l}"] = np.int64(atlas.coordinates_for_alignment(s, l).shape[0])
    pairs = [(0, 1), (0, 2), (1, 2)]
    for (a, b) in pairs:
        out[f"intersect_{a}_{b}"] = np.bool_(bool(atlas.check_submap_intersection(a, b)))
        for l in range(c["n_levels"]):
            for lt in ("L2", "L1"):
                for p in atlas.parameters():
                    p.grad = None
                d = miso.pairwise_loss_latent(atlas, None, a, b, level=l, fdim=c["fdim"],
                                              align_loss=lt, device="cpu")
                (val,) = d.values()
                key = f"latent_{a}_{b}_l{l}_{lt}"
                out[key] = np.float64(val.item())
                if val.requires_grad:
                    val.backward()
                    for which, s in (("src", a), ("dst", b)):
                        gr = atlas.rotation_corrections[s].grad
                        gt_ = atlas.translation_corrections[s].grad
                        out[key + f"_gR_{which}"] = (gr if gr is not None else torch.zeros(1, 3)).numpy().copy()
                        out[key + f"_gt_{which}"] = (gt_ if gt_ is not None else torch.zeros(3, 1)).numpy().copy()
    # 3 (+1: upstream loops num_iters+1 times) Adam iterations of multi-submap alignment
    for p in atlas.parameters():
        p.grad = None
    for s in range(c["n_submaps"]):
        atlas.get_submap(s).lock_feature()
    for l in range(c["n_levels"]):
        loss_tuple = (f"latent{l}", lambda at, ld, a, b, _l=l: miso.pairwise_loss_latent(
            at, ld, a, b, level=_l, fdim=c["fdim"], align_loss="L2", device="cpu"))

        class _DS(torch.utils.data.Dataset):
            def __len__(self):
                return 1

            def __getitem__(self, i):
                return 0

        rbase.generic_align_multiple_submaps(atlas, _DS(), loss_tuple, num_iters=3, lr=1e-2,
                                             verbose=False)
        out[f"align_l{l}_dr"] = np.stack([p.detach().numpy().copy() for p in atlas.rotation_corrections])
        out[f"align_l{l}_dt"] = np.stack([p.detach().numpy().copy() for p in atlas.translation_corrections])
    np.savez_compressed(gc.golden_path("atlas"), **out)
    print("[atlas]", {k: (v.item() if np.ndim(v) == 0 else v.shape) for k, v in out.items()
                      if k.startswith("latent") and k.endswith(("L2", "L1"))})


def gen_atlas_branches(GridAtlas, miso, gc):
    """The optional branches of pairwise_loss_latent (grid_opt/align/miso.py:147-180, :204-209): stability pruning,
    truncation pruning, seeded subsampling, the cos and InfoNCE losses -- value and pose gradients of one pair."""
    atlas = build_atlas(GridAtlas, gc)
    c, b = gc.ATLAS, gc.ATLAS_BRANCHES
    for s in range(c["n_submaps"]):
        net = atlas.get_submap(s)
        with torch.no_grad():
            for l, f in enumerate(gc.atlas_stability(s)):
                net.feature_stability[l].feature.copy_(T(f))
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    src, dst = b["pair"]
    out = {}
    variants = [("stab_l0", dict(level=0, align_loss="L2", stability_thresh=b["stability_thresh"])),
                ("stab_l1", dict(level=1, align_loss="L1", stability_thresh=b["stability_thresh"])),
                ("trunc_l1", dict(level=1, align_loss="L2", trunc_factor=b["trunc_factor"])),
                ("trunc_stab_l0", dict(level=0, align_loss="L1", trunc_factor=4.0, stability_thresh=0.2)),
                ("cos_l0", dict(level=0, align_loss="cos")),
                ("cos_l1", dict(level=1, align_loss="cos")),
                ("nce_l0", dict(level=0, align_loss="InfoNCE")),
                ("sub_l1", dict(level=1, align_loss="L2", subsample_points=b["subsample_points"])),
                ("sub_cos_l1", dict(level=1, align_loss="cos", subsample_points=b["subsample_points"]))]
    for name, kw in variants:
        for p in atlas.parameters():
            p.grad = None
        if "subsample_points" in kw:
            np.random.seed(b["subsample_seed"])
            n_all = atlas.coordinates_for_alignment(src, kw["level"]).shape[0]
            out[name + "_draw"] = np.random.choice(n_all, min(kw["subsample_points"], n_all), replace=False)
            np.random.seed(b["subsample_seed"])          # the call below makes the same draw
        d = miso.pairwise_loss_latent(atlas, None, src, dst, fdim=c["fdim"], device="cpu", **kw)
        (val,) = d.values()
        out[name] = np.float64(val.item())
        val.backward()
        for which, s in (("src", src), ("dst", dst)):
            out[f"{name}_gR_{which}"] = atlas.rotation_corrections[s].grad.numpy().copy()
            out[f"{name}_gt_{which}"] = atlas.translation_corrections[s].grad.numpy().copy()
    np.savez_compressed(gc.golden_path("atlas_branches"), **out)
    print("[atlas_branches]", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


def gen_losses(GridNet, rloss, risdf, gc):
    """MisoLossMapping / MisoLossTracking on a 3-keyframe GridNet + iSDF helpers."""
    case = dict(gc.CASES["small"])
    K = 3
    net = build_gridnet(GridNet, gc, case, num_poses=K, optimize_pose=True)
    rs = np.random.RandomState(99)
    for k in range(K):
        Rk = T(gc.rodrigues(rs.uniform(-0.2, 0.2, 3)).astype(np.float32))
        tk = T(rs.uniform(-0.1, 0.1, (3, 1)).astype(np.float32))
        net.set_initial_kf_pose(k, Rk, tk, kf_key=f"KF{k}")
    with torch.no_grad():
        net.rotation_corrections.copy_(T(rs.uniform(-0.05, 0.05, (K, 3)).astype(np.float32)))
        net.translation_corrections.copy_(T(rs.uniform(-0.05, 0.05, (K, 3, 1)).astype(np.float32)))
    net.unlock_feature()
    net.unlock_pose()
    pts = gc.make_points(case)
    n = pts.shape[0]
    ids = rs.randint(0, K, size=(n, 1)).astype(np.int64)
    sdf_t, valid, sign, weight = gc.make_targets(case, n)
    model_input = {"coords_frame": T(pts)[None], "sample_frame_ids": T(ids)[None],
                   "weights": T(weight)[None]}
    gt = {"sdf": T(sdf_t)[None], "sdf_valid": T(valid)[None], "sdf_signs": T(sign)[None]}
    out = dict(frame_ids=ids, kf_R=net.Rwk.numpy().copy(), kf_t=net.twk.numpy().copy(),
               kf_dr=net.rotation_corrections.detach().numpy().copy(),
               kf_dt=net.translation_corrections.detach().numpy().copy())
    for lt in ("L1", "L2"):
        lossf = rloss.MisoLossMapping(loss_type=lt, weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1,
                                      trunc_dist=0.15)
        for p in net.parameters():
            p.grad = None
        d = lossf.compute(net, model_input, gt)
        tot = sum(v.mean() for v in d.values())
        tot.backward()
        for k_, v in d.items():
            out[f"map_{lt}_{k_}"] = np.float64(v.item())
        out[f"map_{lt}_gdr"] = net.rotation_corrections.grad.numpy().copy()
        out[f"map_{lt}_gdt"] = net.translation_corrections.grad.numpy().copy()
        for l in range(case["n_levels"]):
            out[f"map_{lt}_gfeat{l}"] = net.features[l].feature.grad.numpy().copy()
    gt_valid_all = dict(gt)
    gt_valid_all["sdf_valid"] = torch.ones_like(gt["sdf_valid"])
    for lt in ("L1", "L2", "GM"):
        lossf = rloss.MisoLossTracking(weight_sdf=1.0, loss_type=lt, trunc_dist=0.12, gm_scale_sdf=0.1)
        for p in net.parameters():
            p.grad = None
        d = lossf.compute(net, model_input, gt_valid_all)
        tot = sum(v.mean() for v in d.values())
        tot.backward()
        for k_, v in d.items():
            out[f"track_{lt}_{k_}"] = np.float64(v.item())
        out[f"track_{lt}_gdr"] = net.rotation_corrections.grad.numpy().copy()
        out[f"track_{lt}_gdt"] = net.translation_corrections.grad.numpy().copy()
    # iSDF helpers on fixed random inputs (loss_isdf.py:280-365)
    sdf_p = T((0.2 * rs.standard_normal((1, n, 1))).astype(np.float32))
    bounds = T(np.abs(0.3 * rs.standard_normal((1, n, 1))).astype(np.float32))
    for lt in ("L1", "L2"):
        mat, fsix = risdf.sdf_loss(sdf_p.clone(), bounds, 0.15, loss_type=lt)
        tot, tot_mat, _ = risdf.tot_loss(mat, None, None, fsix, bounds, 0.1, 5.0, 0.0, 0.0)
        out[f"isdf_{lt}_total"] = np.float64(tot.item())
        out[f"isdf_{lt}_mat"] = tot_mat.numpy().copy()
    out["isdf_sdf"] = sdf_p.numpy()
    out["isdf_bounds"] = bounds.numpy()
    # iSDFLoss.compute_slam on the GridNet (loss_isdf.py:46-93)
    il = risdf.iSDFLoss("grid_net", trunc_weight=5.0, trunc_distance=0.15, loss_type="L1",
                        slam_mode=True)
    for p in net.parameters():
        p.grad = None
    gt_b = {"sdf": T(np.abs(sdf_t))[None]}
    d = il.compute(net, model_input, gt_b)
    d["sdf"].backward()
    out["isdf_slam_sdf"] = np.float64(d["sdf"].item())
    out["isdf_slam_gdr"] = net.rotation_corrections.grad.numpy().copy()
    for l in range(case["n_levels"]):
        out[f"isdf_slam_gfeat{l}"] = net.features[l].feature.grad.numpy().copy()
    np.savez_compressed(gc.golden_path("losses"), **out)
    print("[losses]", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


def gen_trainer(GridNet, rloss, rtrainer, gc):
    """GridTrainer: 6 epochs, coordinate+joint, max_epochs_in_level=2 -> features."""
    case = dict(gc.CASES["small"])
    rs = np.random.RandomState(123)
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t, valid, sign, weight = gc.make_targets(case, n)

    class _DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            mi = {"coords_frame": T(pts), "sample_frame_ids": torch.zeros(n, 1, dtype=torch.int64),
                  "weights": T(weight)}
            g = {"sdf": T(sdf_t), "sdf_valid": T(valid), "sdf_signs": T(sign)}
            return mi, g

    out = {}
    for mode in ("joint", "coordinate+joint"):
        net = build_gridnet(GridNet, gc, case, num_poses=1, optimize_pose=False, stability=True)
        net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
        net.unlock_feature()
        net.lock_pose()
        cfg_train = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 6,
                     "ckpt_every": -1, "eval_every": -1, "eval_metric": None,
                     "pretrained_model": None, "log_dir": "/tmp/miso_golden_log",
                     "relchange_tol": 0, "max_epochs_in_level": 2, "grid_training_mode": mode}
        lossf = rloss.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1,
                                      trunc_dist=0.15)
        loader = torch.utils.data.DataLoader(_DS(), batch_size=1, shuffle=False, num_workers=0)

        class _Writer:  # harness patch: tensorboard is absent from the image
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

        rtrainer.SummaryWriter = _Writer
        tr = rtrainer.GridTrainer(cfg_train, net, lossf, loader, None, "cpu", torch.float32)
        tr.train()
        tag = mode.replace("+", "_")
        for l in range(case["n_levels"]):
            out[f"{tag}_feat{l}"] = net.features[l].feature.detach().numpy().copy()
            out[f"{tag}_stab{l}"] = net.feature_stability[l].feature.detach().numpy().copy()
    np.savez_compressed(gc.golden_path("trainer"), **out)
    print("[trainer] ok", {k: float(np.abs(v).sum()) for k, v in out.items()})


def gen_tracker(GridNet, rtracker, gc):
    """Tracker.lm_step (tracker.py:148-212) on a one-KF synthetic batch."""
    case = dict(gc.CASES["small"])
    net = build_gridnet(GridNet, gc, case, num_poses=2, optimize_pose=True)
    rs = np.random.RandomState(321)
    R0 = T(gc.rodrigues(rs.uniform(-0.1, 0.1, 3)).astype(np.float32))
    t0 = T(rs.uniform(-0.05, 0.05, (3, 1)).astype(np.float32))
    net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
    net.set_initial_kf_pose(1, R0, t0, kf_key="KF1")
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t = (0.05 * rs.standard_normal((n, 1))).astype(np.float32)

    class _DS(torch.utils.data.Dataset):
        def select_keyframes(self, kfs):
            pass

        def __len__(self):
            return 1

        def __getitem__(self, i):
            mi = {"coords_frame": T(pts), "sample_frame_ids": torch.ones(n, 1, dtype=torch.int64),
                  "weights": torch.ones(n, 1)}
            g = {"sdf": T(sdf_t), "sdf_valid": torch.ones(n, 1), "sdf_signs": torch.zeros(n, 1)}
            return mi, g

    out = dict(R0=R0.numpy(), t0=t0.numpy(), sdf=sdf_t)
    for lt in ("GM", "L2"):
        with torch.no_grad():
            net.rotation_corrections.zero_()
            net.translation_corrections.zero_()
        cfg = {"device": "cpu", "train": {},
               "tracking": {"learning_rate": 1e-3, "verbose": False, "gm_scale_sdf": 0.1,
                            "lm_lambda": 5.0, "lm_max_iter": 3, "lm_tol_deg": 0.0, "lm_tol_m": 0.0,
                            "loss_type": lt, "trunc_dist": 0.12, "solver": "lm"}}
        trk = rtracker.Tracker.__new__(rtracker.Tracker)
        # bypass the ctor's isinstance(SubmapDataset) plumbing: set the fields lm_step reads
        trk.grid = net
        trk.dataset = _DS()
        trk.train_loader = torch.utils.data.DataLoader(trk.dataset, shuffle=False, batch_size=1)
        trk.cfg = cfg
        for k_, v in cfg["tracking"].items():
            setattr(trk, k_, v)
        trk.lr = 1e-3
        infos = []
        for it in range(3):
            info = trk.lm_step(1)
            infos.append([info["delta_R_deg"], info["delta_t_norm"], info["grad_norm"], info["fov_overlap"]])
            out[f"lm_{lt}_dr_{it}"] = net.rotation_corrections.detach().numpy().copy()
            out[f"lm_{lt}_dt_{it}"] = net.translation_corrections.detach().numpy().copy()
        out[f"lm_{lt}_info"] = np.array(infos, dtype=np.float64)
    np.savez_compressed(gc.golden_path("tracker"), **out)
    print("[tracker]", out["lm_GM_info"])


def gen_so3(rgeom, gc):
    """so3_exp_map / apply_pose_correction values + grads (restated shim => the
    pytorch3d boundary stays 'parity unpinned'; this pins OUR restatement)."""
    out = {}
    for i, w in enumerate([[0.0, 0.0, 0.0], [1e-3, -2e-3, 5e-4], [0.3, -0.2, 0.1], [1.2, 0.4, -0.9]]):
        dr = torch.tensor([w], dtype=torch.float32, requires_grad=True)
        dt = torch.tensor([[0.1], [-0.2], [0.3]], dtype=torch.float32, requires_grad=True)
        R0 = T(gc.rodrigues([0.2, 0.1, -0.3]).astype(np.float32))
        t0 = torch.tensor([[1.0], [2.0], [3.0]])
        R, t = rgeom.apply_pose_correction(R0, t0, dr, dt)
        wgt = torch.arange(9, dtype=torch.float32).reshape(3, 3) / 10
        (R * wgt).sum().backward()
        out[f"R_{i}"] = R.detach().numpy()
        out[f"t_{i}"] = t.detach().numpy()
        out[f"gdr_{i}"] = dr.grad.numpy().copy()
    np.savez_compressed(gc.golden_path("so3"), **out)
    print("[so3] ok")


def gen_samples(gc):
    """Sample generation: the reference's own PosedSdfRgbd.getitem_sdf and PosedSdf3DLidar.sample_frames /
    getitem_world, driven on in-memory frames (the constructors only add file IO).  The random draws the
    reference makes are recorded by replaying its RNG calls from the same seed."""
    import grid_opt.datasets.sdf_rgbd as rr
    import grid_opt.datasets.sdf_3d_lidar as rl
    import grid_opt.utils.utils_data as rdata
    out = {}
    c = gc.RGBD
    inp = gc.rgbd_inputs()
    ds = object.__new__(rr.PosedSdfRgbd)
    ds._cam_params = rdata.CameraParameters(fx=c["fx"], fy=c["fy"], cx=c["cx"], cy=c["cy"], H=c["H"], W=c["W"])
    ds.device = "cpu"
    ds.dirs_C = rr.ray_dirs_C(1, c["H"], c["W"], c["fx"], c["fy"], c["cx"], c["cy"], "cpu", depth_type="z")
    ds.min_depth, ds.max_depth, ds.voxel_size = c["min_depth"], 12.0, None
    ds.n_rays, ds.dist_behind_surf = c["n_rays"], c["dist_behind_surf"]
    ds.n_strat_samples, ds.n_surf_samples, ds.trunc_dist = c["n_strat"], c["n_surf"], c["trunc_dist"]
    ds.bounds_method, ds.normal_trunc_dist, ds.use_clip = "ray", 0.30, False
    ds._num_frames = c["n_frames"]
    ds._depth_batch, ds._T_WC_batch, ds._norm_batch = T(inp["depth"]), T(inp["T_WC"]), T(inp["normals"])
    ds.R_world_frame_gt, ds.t_world_frame_gt = T(inp["R"]), T(inp["t"])
    for tag, sel in (("all", None), ("sel", c["selected"])):
        ds._selected_kfs = sel
        nf = c["n_frames"] if sel is None else len(sel)
        # replay of the reference's RNG calls: sample_pixels (utils_sample.py:133-134), stratified_sample (:241),
        # sample_along_rays (:284-286)
        torch.manual_seed(100 + nf)
        total = c["n_rays"] * nf
        ph = torch.randint(0, c["H"], (total,))
        pw = torch.randint(0, c["W"], (total,))
        pb = torch.arange(nf).repeat_interleave(c["n_rays"])
        idx = list(range(c["n_frames"])) if sel is None else sel
        d = ds._depth_batch[idx][pb, ph, pw]
        ok = (d != 0) & ~torch.isnan(ds._norm_batch[idx][pb, ph, pw, 0])
        n1 = int(ok.sum())
        u = torch.rand(n1, c["n_strat"])
        g = torch.normal(torch.zeros(n1, c["n_surf"] - 1), 0.1)
        torch.manual_seed(100 + nf)
        inputs, gt = ds.getitem_sdf(0)
        out.update({f"rgbd_{tag}_pix_h": ph.numpy(), f"rgbd_{tag}_pix_w": pw.numpy(), f"rgbd_{tag}_u": u.numpy(),
                    f"rgbd_{tag}_g": g.numpy(), f"rgbd_{tag}_coords": inputs["coords_frame"].numpy(),
                    f"rgbd_{tag}_ids": inputs["sample_frame_ids"].numpy(),
                    f"rgbd_{tag}_weights": inputs["weights"].numpy(), f"rgbd_{tag}_sdf": gt["sdf"].numpy(),
                    f"rgbd_{tag}_valid": gt["sdf_valid"].numpy(), f"rgbd_{tag}_signs": gt["sdf_signs"].numpy()})
        print(f"[samples] rgbd {tag}: rays {total} -> first filter {n1} -> rows {tuple(gt['sdf'].shape)}")

    # normals as load_rgbd estimates them (sdf_rgbd.py:205-207)
    for f in range(c["n_frames"]):
        pc = rr.pointcloud_from_depth_torch(T(inp["depth"][f]), c["fx"], c["fy"], c["cx"], c["cy"])
        out[f"rgbd_pc_{f}"] = pc.numpy()
        out[f"rgbd_est_normals_{f}"] = rr.estimate_pointcloud_normals(pc).numpy()

    lc = gc.LIDAR
    frames = gc.lidar_inputs()
    dl = object.__new__(rl.PosedSdf3DLidar)
    for k in ("frame_batchsize", "frame_samples", "near_surface_n", "near_surface_std", "free_space_n",
              "behind_surface_n", "trunc_dist", "min_dist_ratio", "max_range"):
        setattr(dl, k, lc[k])
    dl.max_range_hehind_surface = 4 * lc["near_surface_std"]
    dl.distance_std = 0.0
    dl._num_frames = lc["n_frames"]
    dl.R_world_frame_gt = torch.stack([T(f["R"]) for f in frames])
    dl.t_world_frame_gt = torch.stack([T(f["t"]) for f in frames])
    dl.frames_lidar = [{"points_global": f["points_global"]} for f in frames]
    dl.frames_data = []
    rl.tqdm = lambda it, **k: it
    # replay of the numpy RNG calls of sample_frames (:235,:257,:277,:295) per frame
    np.random.seed(77)
    for f, fr in enumerate(frames):
        n = fr["points_global"].shape[0]
        keep = min(lc["frame_samples"], n)
        out[f"lidar_perm_{f}"] = np.random.permutation(n)[:keep]
        out[f"lidar_g_near_{f}"] = np.random.randn(keep * lc["near_surface_n"], 1)
        out[f"lidar_u_free_{f}"] = np.random.rand(keep * lc["free_space_n"], 1)
        out[f"lidar_u_behind_{f}"] = np.random.rand(keep * lc["behind_surface_n"], 1)
    np.random.seed(77)
    dl.sample_frames()
    for f, fd in enumerate(dl.frames_data):
        for k, v in fd.items():
            out[f"lidar_{k}_{f}"] = v.numpy()
    dl._selected_kfs = None
    np.random.seed(78)
    sel = [np.random.choice(fd["points_frame"].shape[0], size=min(lc["frame_batchsize"], fd["points_frame"].shape[0]),
                            replace=False) for fd in dl.frames_data]
    np.random.seed(78)
    inputs, gt = dl.getitem_world(0)
    for f, sidx in enumerate(sel):
        out[f"lidar_choice_{f}"] = sidx
    out.update({"lidar_batch_coords": inputs["coords_frame"].numpy(), "lidar_batch_ids": inputs["sample_frame_ids"].numpy(),
                "lidar_batch_weights": inputs["weights"].numpy(), "lidar_batch_sdf": gt["sdf"].numpy(),
                "lidar_batch_valid": gt["sdf_valid"].numpy(), "lidar_batch_signs": gt["sdf_signs"].numpy()})
    print("[samples] lidar batch", tuple(gt["sdf"].shape))
    np.savez_compressed(gc.golden_path("samples"), **out)


# --------------------------------------------------------------------------- #
def build_atlas_two_kf(GridAtlas, gc):
    """build_atlas with a second, non-identity keyframe in every submap (global keyframe ids 2s, 2s+1)."""
    c = gc.ATLAS
    cfg = gc.model_cfg(c["bound"], c["base_cell"], c["scale"], c["n_levels"], c["fdim"], c["hidden"])
    atlas = GridAtlas(cfg, device="cpu")
    dec = {k: T(v) for k, v in gc.make_decoder(c).items()}
    for s, sub in enumerate(gc.atlas_inputs()):
        atlas.add_submap(torch.tensor(c["bound"], dtype=torch.float32), T(sub["R"]), T(sub["t"]), num_poses=2)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
        R2, t2 = gc.atlas_second_kf_pose(s)
        atlas.add_kf(T(R2), T(t2))
        net = atlas.get_submap(s)
        with torch.no_grad():
            for l, f in enumerate(sub["features"]):
                net.features[l].feature.copy_(T(f))
        net.decoder.load_state_dict(dec)
        atlas.set_submap_pose_correction(s, T(sub["dr"]), T(sub["dt"]))
    return atlas


class _OneBatch(torch.utils.data.Dataset):
    def __init__(self, mi, g):
        self.mi, self.g = mi, g

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return ({k: T(v[0]) for k, v in self.mi.items()}, {k: T(v[0]) for k, v in self.g.items()})


def gen_extra(GridNet, GridAtlas, miso, rtrainer, gc):
    """Rows without a golden so far: grid_pool_3d_avg, pairwise_loss_sdf (SDF-space pair loss through the per-keyframe
    loop), local_opt.optimize_grid_net / optimize_grid_atlas (iSDF / iSDFSubmap losses through GridTrainer)."""
    import grid_opt.utils.utils as ru
    import grid_opt.local_opt as rlocal
    out = {}
    # -- scatter-average pooling (utils.py:239-291)
    pts, feats = gc.pool_inputs()
    out["pool"] = ru.grid_pool_3d_avg(T(pts), T(feats), torch.tensor(gc.POOL["bound"], dtype=torch.float32),
                                      gc.POOL["cell"]).numpy()
    # -- pairwise_loss_sdf (align/miso.py:14-113)
    atlas = build_atlas_two_kf(GridAtlas, gc)
    mi, g = gc.atlas_sdf_batch()
    loader = torch.utils.data.DataLoader(_OneBatch(mi, g), batch_size=1, shuffle=False, num_workers=0)
    for (a, b) in [(0, 1), (1, 2), (2, 0)]:
        for lt in ("L2", "L1", "GM"):
            for p in atlas.parameters():
                p.grad = None
            d = miso.pairwise_loss_sdf(atlas, loader, a, b, align_loss=lt, device="cpu")
            (val,) = d.values()
            key = f"pairsdf_{a}_{b}_{lt}"
            assert list(d) == [f"align_sdf_{a}_{b}"]
            out[key] = np.float64(val.item())
            val.backward()
            for which, s in (("src", a), ("dst", b)):
                out[key + f"_gR_{which}"] = atlas.rotation_corrections[s].grad.numpy().copy()
                out[key + f"_gt_{which}"] = atlas.translation_corrections[s].grad.numpy().copy()

    class _Writer:  # harness patch: tensorboard is absent from the image
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    rtrainer.SummaryWriter = _Writer
    # -- local_opt.optimize_grid_net with the iSDF loss (local_opt.py:60-103)
    case = dict(gc.CASES["small"])
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t = gc.make_targets(case, n)[0]
    cfg = gc.local_opt_cfg("iSDF")
    net = build_gridnet(GridNet, gc, case, num_poses=1, optimize_pose=False, stability=True)
    net.unlock_feature()
    net.lock_pose()
    ds = _OneBatch({"coords": pts[None], "normals": np.zeros((1, n, 3), np.float32)},
                   {"sdf": sdf_t[None], "grad_vec": np.zeros((1, n, 3), np.float32)})
    net, info = rlocal.optimize_grid_net(net, ds, cfg, iterations=5, learning_rate=2e-3, train_mode="joint",
                                         iterations_per_level=2)
    for l in range(case["n_levels"]):
        out[f"localopt_net_feat{l}"] = net.features[l].feature.detach().numpy().copy()
    out["localopt_net_epochs"] = np.asarray(info["trainer_epoch"], dtype=np.int64)
    out["localopt_net_loss"] = np.asarray(info["trainer_total_loss"], dtype=np.float64)
    # -- local_opt.optimize_grid_atlas with the iSDFSubmap loss (local_opt.py:127-154)
    atlas = build_atlas_two_kf(GridAtlas, gc)
    mi, g = gc.atlas_sdf_batch()
    nrow = mi["coords_frame"].shape[1]
    owner = (mi["sample_frame_ids"] // 2).astype(np.int64)
    ds = _OneBatch({"coords_submap": mi["coords_frame"], "submap_idxs": owner}, {"sdf": g["sdf"], "sdf_valid": g["sdf_valid"]})
    atlas, info = rlocal.optimize_grid_atlas(atlas, ds, gc.local_opt_cfg("iSDFSubmap"), iterations=4, learning_rate=2e-3,
                                             train_mode="coordinate")
    for s in range(gc.ATLAS["n_submaps"]):
        for l in range(gc.ATLAS["n_levels"]):
            out[f"localopt_atlas_s{s}_feat{l}"] = atlas.get_submap(s).features[l].feature.detach().numpy().copy()
    assert info == {}
    out["localopt_atlas_dr"] = np.stack([p.detach().numpy().copy() for p in atlas.rotation_corrections])
    out["localopt_atlas_dt"] = np.stack([p.detach().numpy().copy() for p in atlas.translation_corrections])
    np.savez_compressed(gc.golden_path("extra"), **out)
    print("[extra] ok", {k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


def gen_geometry(rgeom, gc):
    """utils_geometry helpers that run without pytorch3d's un-shimmed functions (so3_relative_angle and
    matrix_to_axis_angle are not restated in tools/ref_shims, so rotation_rmse / get_pose_correction are not here)."""
    import tempfile
    g = gc.geometry_inputs()
    out = {}
    out["batch_world"] = rgeom.batch_transform_to_world_frame(T(g["pts"]), T(g["spans"]), T(g["R"]), T(g["t"]), T(g["dr"]),
                                                              T(g["dt"])).numpy()
    Rf, tf = rgeom.transform_poses_from(T(g["R"]), T(g["t"]), T(g["R"][1]), T(g["t"][1]))
    out["poses_from_R"], out["poses_from_t"] = Rf.numpy(), tf.numpy()
    out["aabb"] = rgeom.aabb_torch(T(g["cloud"]), buffer=0.25).numpy()
    for vs in (0.5, 2.0):
        out[f"voxel_{vs}"] = rgeom.voxel_down_sample_torch(T(g["cloud"]), vs).numpy()
    p, s = rgeom.crop_points(T(g["cloud"]), T(g["stamps"]), min_z_th=-1.0, max_z_th=2.0, min_range=2.75, max_range=9.0)
    out["crop_pts"], out["crop_ts"] = p.numpy(), s.numpy()
    out["t_rmse"] = np.float64(rgeom.translation_rmse(T(g["t"]), T(g["dt"])))
    out["t_mean"] = np.float64(rgeom.translation_mean_error(T(g["t"]), T(g["dt"])))
    out["chordal_deg"] = np.float64(rgeom.chordal_to_degree(0.7))
    np.random.seed(7)
    out["gauss_t"] = rgeom.gaussian_translations(5, 0.5).numpy()
    out["uniform_t"] = rgeom.uniform_translations(5, np.array([[-1.0, 1.0], [0.0, 2.0], [3.0, 4.0]])).numpy()
    out["fixed_len_t"] = rgeom.fixed_length_translations(5, 0.3).numpy()
    out["wrapped_R"] = rgeom.wrapped_gaussian_rotations(5, std_rad=0.2).numpy()
    out["fixed_angle_R"] = rgeom.fixed_angle_rotations(5, 0.4).numpy()
    poses = np.tile(np.eye(4), (4, 1, 1))
    poses[:, :3, :3], poses[:, :3, 3:] = g["R"], g["t"]
    with tempfile.TemporaryDirectory() as d:
        rgeom.write_kitti_format_poses(os.path.join(d, "traj"), poses)
        out["kitti_text"] = np.frombuffer(open(os.path.join(d, "traj_kitti.txt"), "rb").read(), dtype=np.uint8)
        out["kitti_read"] = np.stack(rgeom.read_kitti_format_poses(os.path.join(d, "traj_kitti.txt")))
    out["pose_ok"] = np.array([rgeom.check_numpy_pose_matrix(poses[0]), rgeom.check_numpy_pose_matrix(poses[0] * 1.01),
                               rgeom.check_numpy_pose_matrix(np.full((4, 4), np.nan))])
    np.savez_compressed(gc.golden_path("geometry"), **out)
    print("[geometry] ok", {k: v.shape for k, v in out.items()})


def gen_formats(GridNet, GridAtlas, rloss, rtrainer, gc):
    """On-disk formats written BY THE REFERENCE (data only: tensors, plain containers and, for the whole-module
    pickle, dotted class names): a Trainer.save_model checkpoint (trainer.py:319-332) and torch.save(grid_atlas)
    as demo/build_submaps.py:141 writes it.  tests/ loads them through miso_amd.compat."""
    import shutil
    import tempfile
    case = dict(gc.CASES["small"])
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t, valid, sign, weight = gc.make_targets(case, n)
    net = build_gridnet(GridNet, gc, case, num_poses=1, optimize_pose=False, stability=True)
    net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
    net.unlock_feature()
    net.lock_pose()
    ds = _OneBatch({"coords_frame": pts[None], "sample_frame_ids": np.zeros((1, n, 1), np.int64), "weights": weight[None]},
                   {"sdf": sdf_t[None], "sdf_valid": valid[None], "sdf_signs": sign[None]})
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)

    class _Writer:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    rtrainer.SummaryWriter = _Writer
    d = tempfile.mkdtemp()
    cfg_train = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 3, "ckpt_every": -1,
                 "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": d, "relchange_tol": 0,
                 "max_epochs_in_level": 2, "grid_training_mode": "joint"}
    lossf = rloss.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
    tr = rtrainer.GridTrainer(cfg_train, net, lossf, loader, None, "cpu", torch.float32)
    tr.train()
    tr.save_model(3, "ref_checkpoint")
    shutil.copy(os.path.join(tr.ckpt_dir, "ref_checkpoint.pt"), os.path.join(gc.GOLDEN_DIR, "ref_checkpoint.pt"))
    xq = T(pts[:256])
    out = {"ckpt_forward": net(xq).detach().numpy(), "ckpt_keys": np.array(sorted(net.state_dict().keys()))}
    # whole-module pickle of a two-submap atlas (small bound so that the fixture stays small)
    c = dict(gc.ATLAS, bound=[[-1.0, 1.0], [-0.5, 0.5], [-1.0, 1.0]])
    cfg = gc.model_cfg(c["bound"], c["base_cell"], c["scale"], c["n_levels"], c["fdim"], c["hidden"])
    atlas = GridAtlas(cfg, device="cpu")
    dec = {k: T(v) for k, v in gc.make_decoder(c).items()}
    rs = np.random.RandomState(17)
    for s_ in range(2):
        atlas.add_submap(torch.tensor(c["bound"], dtype=torch.float32), T(gc.rodrigues([0.0, 0.1 * s_, 0.0]).astype(np.float32)),
                         torch.tensor([[0.9 * s_], [0.0], [0.1 * s_]]), num_poses=2)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
        atlas.add_kf(T(gc.rodrigues([0.05, 0.0, 0.1]).astype(np.float32)), torch.tensor([[0.1], [0.05], [0.0]]))
        sub = atlas.get_submap(s_)
        with torch.no_grad():
            for f in sub.features:
                f.feature.copy_(T((rs.standard_normal(tuple(f.feature.shape)) * 0.1).astype(np.float32)))
        sub.decoder.load_state_dict(dec)
        atlas.set_submap_pose_correction(s_, T(rs.uniform(-0.05, 0.05, (1, 3)).astype(np.float32)),
                                         T(rs.uniform(-0.1, 0.1, (3, 1)).astype(np.float32)))
    torch.save(atlas, os.path.join(gc.GOLDEN_DIR, "ref_atlas.pth"))
    xw = T(rs.uniform(-1.0, 1.8, (300, 3)).astype(np.float32))
    out["atlas_x"] = xw.numpy()
    out["atlas_forward"] = atlas(xw).detach().numpy()
    out["atlas_query_feature"] = atlas.query_feature(xw).detach().numpy()
    R1, t1 = atlas.updated_kf_pose_in_world(3)
    out["atlas_kf3_R"], out["atlas_kf3_t"] = R1.detach().numpy(), t1.detach().numpy()
    out["atlas_anchor"] = np.array([atlas.anchor_kf_for_submap(0), atlas.anchor_kf_for_submap(1)])
    np.savez_compressed(gc.golden_path("formats"), **out)
    print("[formats] ok", {k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(gc.GOLDEN_DIR, "ref_atlas.pth")),
          os.path.getsize(os.path.join(gc.GOLDEN_DIR, "ref_checkpoint.pt")))


def gen_encoder(GridNet, gc):
    """Learned initialisation (grid_opt/models/encoder.py, modules.FeaturePrediction, local_opt.initialize_grid_net)
    with seeded random predictor weights (upstream's pretrained ones are not shipped).  Harness patch: the reference's
    query goes through the CUDA extension module `cuda_gridsample` (utils.py:116-118), absent here; its first-order
    semantics are F.grid_sample's (cuda_gridsample.py:84), which stands in -- the eikonal / smoothness terms that need
    the second order stay off."""
    import types
    import torch.nn.functional as F
    cu = types.ModuleType("cuda_gridsample")
    cu.grid_sample_3d = lambda input, grid, padding_mode="zeros", align_corners=True: F.grid_sample(
        input, grid, padding_mode=padding_mode, align_corners=align_corners)
    sys.modules["cuda_gridsample"] = cu
    import grid_opt.models.encoder as renc
    import grid_opt.models.modules as rmod
    import grid_opt.local_opt as rlocal
    # harness patch: ConvInterp builds its convolutions on 'cuda:0' by default before FeaturePrediction moves them
    d = list(rmod.ConvInterp.__init__.__defaults__)
    rmod.ConvInterp.__init__.__defaults__ = tuple("cpu" if v == "cuda:0" else v for v in d)
    case = dict(gc.CASES["small"])
    cfg = {"device": "cpu", "model": gc.model_cfg(case["bound"], case["base_cell"], case["scale"], case["n_levels"],
                                                  case["fdim"], case["hidden"])}
    torch.manual_seed(5)
    enc = renc.Encoder(cfg)
    out = {}
    for l, fe in enumerate(enc.feature_encoders):
        for k, v in fe.state_dict().items():
            out[f"enc{l}.{k}"] = v.numpy().copy()
    net = build_gridnet(GridNet, gc, case, num_poses=1, optimize_pose=False)
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t, valid, sign, _ = gc.make_targets(case, n)
    obs = renc.EncoderObservation(coords_world=T(pts), gt_sdf=T(sdf_t), gt_sdf_sign=T(sign), gt_sdf_valid=T(valid))
    mid = enc.register_grid_model(net)
    res = enc.compute_residuals(mid, [torch.zeros_like(f.feature) for f in net.features], obs)
    for k in ("sdf_constraint", "fs_constraint", "fs_upper_constraint", "fs_lower_constraint"):
        out[f"res_{k}"] = res[k].detach().numpy()
    out["enc_inputs_l1"] = enc.compute_encoder_inputs_from_residuals(res, mid, 1).detach().numpy()
    corr = enc.predict_corrections_until_level(mid, 2, obs, pred_std=0, store_corrections=True)
    for l, c in enumerate(corr):
        out[f"corr{l}"] = c.detach().numpy()
    out["stored_until1_l1_abs"] = np.float64(enc.stored_corrections_until_level(mid, 1)[1].abs().sum().item())
    # pre-training loss of level 1 and its gradient to that level's predictor
    enc.lock_all_params()
    enc.unlock_encoder_at_level(1)
    lossf = renc.EncoderPretrainLoss(target_level=1, sdf_weight=3e3, sign_weight=10.0, pred_std=0.0)
    F_ = 2
    spans = np.array([[0, n // 2], [n // 2, n]], dtype=np.int64)
    mi = {"dataset_index": torch.tensor([mid]), "coords_frame": T(pts)[None], "frame_indices": T(spans)[None],
          "R_world_frame": torch.eye(3).repeat(F_, 1, 1)[None], "t_world_frame": torch.zeros(1, F_, 3, 1)}
    g = {"sdf": T(sdf_t)[None], "sdf_valid": T(valid)[None], "sdf_signs": T(sign)[None]}
    # the grid under test needs 2 pose slots for the 2 frames
    net2 = build_gridnet(GridNet, gc, case, num_poses=2, optimize_pose=False)
    mid2 = enc.register_grid_model(net2)
    mi["dataset_index"] = torch.tensor([mid2])
    ld = lossf.compute(enc, mi, g)
    total = sum(v for v in ld.values())
    total.backward()
    out["pretrain_sdf"], out["pretrain_fs"] = np.float64(ld["sdf"].item()), np.float64(ld["free_space"].item())
    for k, p in enc.feature_encoders[1].named_parameters():
        out[f"pretrain_grad.{k}"] = p.grad.numpy().copy()
    # local_opt.initialize_grid_net(init_mode='encode')
    enc3 = renc.Encoder(cfg)
    for l in range(2):
        enc3.feature_encoders[l].load_state_dict(enc.feature_encoders[l].state_dict())
    net3 = build_gridnet(GridNet, gc, case, num_poses=1, optimize_pose=False)
    net3, info = rlocal.initialize_grid_net(net3, "encode", enc3, obs)
    for l in range(2):
        out[f"init_feat{l}"] = net3.features[l].feature.detach().numpy().copy()
    assert set(info) == {"total_encoder_time"}
    np.savez_compressed(gc.golden_path("encoder"), **out)
    print("[encoder] ok", {k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


def gen_second_order(GridNet, rloss, risdf, gc):
    """The branches of the reference's losses that differentiate a spatial gradient obtained with create_graph=True
    (loss_isdf.py:99-150 eik / grad / smooth; loss.py:638-665 miso_loss_eikonal with grad_method 'autograd' and
    'finitediff'), on a GridNet built with second_order_grid_sample=True (grid_modules.py:63-66).

    The loss code that runs here is the REFERENCE's.  Its sampling op for this configuration is the CUDA extension
    third_party/cuda_gridsample_grad2 (cuda_gridsample.grid_sample_3d), which can be neither built nor run in this
    container, and ATen has no double backward for grid_sampler_3d; so a module named ``cuda_gridsample``
    backed by the oracle's any-order restatement (oracle.ref_torch.trilinear_gather: equal to ATen to first order,
    gradgradcheck'ed in fp64, tests/test_oracle_golden.py) stands in for it.  The random perturbation of the
    smoothness term (torch.randn_like, loss_isdf.py:143) is replaced by a recorded draw."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import ref_torch as R
    shim = types.ModuleType("cuda_gridsample")

    def grid_sample_3d(input, grid, padding_mode="zeros", align_corners=True):
        _, do, ho, wo, _ = grid.shape
        out = R.trilinear_gather(input, grid.reshape(-1, 3), align_corners, padding_mode)
        return out.transpose(0, 1).reshape(1, input.shape[1], do, ho, wo)

    shim.grid_sample_3d = grid_sample_3d
    shim.grid_sample_2d = None
    sys.modules["cuda_gridsample"] = shim
    case = dict(gc.CASES["small"])
    cfg = gc.model_cfg(case["bound"], case["base_cell"], case["scale"], case["n_levels"], case["fdim"], case["hidden"],
                       second_order=True)
    net = GridNet(cfg, device="cpu")
    with torch.no_grad():
        for l, f in enumerate(gc.make_features(case)):
            net.features[l].feature.copy_(T(f))
    net.decoder.load_state_dict({k: T(v) for k, v in gc.make_decoder(case).items()})
    net.unlock_feature()
    assert net.features[0].grid_sample_func is grid_sample_3d
    rs = np.random.RandomState(4242)
    n_rays, per_ray = 96, 5
    n = n_rays * per_ray
    b = np.asarray(case["bound"], dtype=np.float32)
    pts = (b[:, 0] + (b[:, 1] - b[:, 0]) * rs.uniform(0.03, 0.97, size=(n, 3))).astype(np.float32)
    unit = lambda a: (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)
    normals = unit(rs.standard_normal((1, n_rays, 3)))
    grad_vec = unit(rs.standard_normal((1, n_rays, per_ray - 1, 3)))
    grad_vec[0, ::7, 1, :] = np.nan                       # rows the reference replaces by the surface normal (:126-127)
    bounds = np.abs(0.2 * rs.standard_normal((1, n, 1))).astype(np.float32)
    noise = rs.standard_normal((1, n, 3)).astype(np.float32)
    out = dict(coords=pts, normals=normals, grad_vec=grad_vec.copy(), bounds=bounds, noise=noise,
               n_rays=np.int64(n_rays), per_ray=np.int64(per_ray))
    real_randn_like = torch.randn_like
    for tag, kw in (("eik", dict(eik_weight=50.0)),
                    ("grad", dict(grad_weight=0.02)),
                    ("all", dict(eik_weight=50.0, grad_weight=0.02, smooth_weight=0.1)),
                    ("allL2", dict(eik_weight=50.0, grad_weight=0.02, smooth_weight=0.1, loss_type="L2"))):
        il = risdf.iSDFLoss("grid_net", trunc_weight=5.0, trunc_distance=0.15, eik_apply_dist=0.1, smooth_std=0.05,
                            slam_mode=False, **kw)
        for p_ in net.parameters():
            p_.grad = None
        mi = {"coords": T(pts)[None].clone(), "normals": T(normals).clone()}
        gt = {"sdf": T(bounds).clone(), "grad_vec": T(grad_vec).clone()}
        torch.randn_like = lambda t, *a, **k: T(noise).clone()
        try:
            d = il.compute(net, mi, gt)
        finally:
            torch.randn_like = real_randn_like
        sum(v.mean() for v in d.values()).backward()
        for k_, v in d.items():
            out[f"isdf_{tag}_{k_}"] = np.float64(v.item())
        for l in range(case["n_levels"]):
            out[f"isdf_{tag}_gfeat{l}"] = net.features[l].feature.grad.numpy().copy()
        out[f"isdf_{tag}_gcoords"] = mi["coords"].grad.numpy().copy()
    # miso_loss_eikonal, both gradient methods (configs/rgbd/scannet.yaml:45-49 ships 'finitediff')
    sdf_t = (0.12 * rs.standard_normal((n, 1))).astype(np.float32)
    out["eik_gt_sdf"] = sdf_t
    for method in ("autograd", "finitediff"):
        for p_ in net.parameters():
            p_.grad = None
        val = rloss.miso_loss_eikonal(model=net, coords_world=T(pts), gt_sdf=T(sdf_t), eik_trunc_dist=0.1,
                                      grad_method=method, finite_diff_eps=1e-2)
        val.backward()
        out[f"eik_{method}"] = np.float64(val.item())
        for l in range(case["n_levels"]):
            out[f"eik_{method}_gfeat{l}"] = net.features[l].feature.grad.numpy().copy()
    np.savez_compressed(gc.golden_path("second_order"), **out)
    print("[second_order]", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


def main():
    import_reference()
    import golden_cases as gc
    from grid_opt.models.grid_net import GridNet
    from grid_opt.models.grid_atlas import GridAtlas
    import grid_opt.loss as rloss
    import grid_opt.loss_isdf as risdf
    import grid_opt.align.miso as miso
    import grid_opt.align.base as rbase
    import grid_opt.trainer as rtrainer
    import grid_opt.slam.tracker as rtracker
    import grid_opt.utils.utils_geometry as rgeom

    os.makedirs(gc.GOLDEN_DIR, exist_ok=True)
    torch.manual_seed(0)
    np.random.seed(0)
    which = sys.argv[1:] or ["small", "cfg1", "cfg2", "atlas", "losses", "trainer", "tracker", "so3", "samples", "extra", "geometry", "formats", "encoder", "second_order", "atlas_branches"]
    for name in which:
        if name in gc.CASES:
            gen_encode_decode(name, GridNet, rloss, gc)
        elif name == "atlas":
            gen_atlas(GridAtlas, miso, rbase, gc)
        elif name == "losses":
            gen_losses(GridNet, rloss, risdf, gc)
        elif name == "trainer":
            gen_trainer(GridNet, rloss, rtrainer, gc)
        elif name == "tracker":
            gen_tracker(GridNet, rtracker, gc)
        elif name == "so3":
            gen_so3(rgeom, gc)
        elif name == "samples":
            gen_samples(gc)
        elif name == "geometry":
            gen_geometry(rgeom, gc)
        elif name == "formats":
            gen_formats(GridNet, GridAtlas, rloss, rtrainer, gc)
        elif name == "encoder":
            gen_encoder(GridNet, gc)
        elif name == "extra":
            gen_extra(GridNet, GridAtlas, miso, rtrainer, gc)
        elif name == "second_order":
            gen_second_order(GridNet, rloss, risdf, gc)
        elif name == "atlas_branches":
            gen_atlas_branches(GridAtlas, miso, gc)


if __name__ == "__main__":
    main()
