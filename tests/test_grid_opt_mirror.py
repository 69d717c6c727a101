"""The host-side mirror of the reference API (miso_amd.grid_opt) against the golden
vectors captured from the reference: GridNet / GridAtlas queries, the Miso and iSDF
losses, the trainer step, pairwise alignment, LM tracking.  Runs twice: on CPU with the
oracle standing in for the HIP operators (host logic only), and on the GPU with the
real library (marked gpu)."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def G(name):
    return np.load(gc.golden_path(name))


def make_gridnet(case, dev, num_poses=1, optimize_pose=False, stability=False):
    from miso_amd.grid_opt.models.grid_net import GridNet
    cfg = gc.model_cfg(case["bound"], case["base_cell"], case["scale"], case["n_levels"], case["fdim"],
                       case["hidden"], num_poses=num_poses, optimize_pose=optimize_pose)
    net = GridNet(cfg, device=dev)
    with torch.no_grad():
        for l, f in enumerate(gc.make_features(case)):
            assert tuple(net.features[l].feature.shape) == f.shape
            net.features[l].feature.copy_(T(f))
        if stability:
            for l, f in enumerate(gc.make_stability(case)):
                net.feature_stability[l].feature.copy_(T(f))
    net.decoder.load_state_dict({k: T(v) for k, v in gc.make_decoder(case).items()})
    return net.to(dev)


def close(a, b, rtol, atol=0.0):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return
    tol = atol + rtol * b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e}"


@pytest.mark.parametrize("name", ["small", "cfg1"])
def test_gridnet_queries(device_backend, name):
    dev = device_backend
    case = gc.CASES[name]
    g = G(name)
    net = make_gridnet(case, dev, stability=True)
    assert net.features[0].feature.is_contiguous(memory_format=torch.channels_last_3d)
    x = T(gc.make_points(case)).to(dev)
    close(net.query_feature(x), T(g["feats"]), 0, 1e-6)
    close(net.query_stability(x), T(g["stab"]), 1e-6, 2e-6)
    close(net(x), T(g["sdf"]), 0, 1e-5)
    close(net.features[0].interpolate(x), T(g["feats"])[:, :case["fdim"]], 0, 1e-6)
    # ignore_level zeroes that level's columns only
    net.ignore_level(0)
    f = net.query_feature(x)
    assert torch.all(f[:, :case["fdim"]] == 0)
    close(f[:, case["fdim"]:], T(g["feats"])[:, case["fdim"]:], 0, 1e-6)
    net.include_level(0)
    # vertex positions: voxel centres, z-major, column 0 = x
    vp = net.features[0].vertex_positions()
    _, _, nz, ny, nx = net.features[0].feature.shape
    assert vp.shape == (nz * ny * nx, 3)
    b = torch.tensor(case["bound"])
    cell = (b[:, 1] - b[:, 0]) / torch.tensor([nx, ny, nz])
    torch.testing.assert_close(vp[0], b[:, 0] + 0.5 * cell, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(vp[1] - vp[0], torch.tensor([cell[0], 0, 0]), rtol=1e-4, atol=1e-6)


def _loss_inputs(case, dev, K=3):
    rs = np.random.RandomState(99)
    net = make_gridnet(case, dev, num_poses=K, optimize_pose=True)
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
    mi = {"coords_frame": T(pts)[None].to(dev), "sample_frame_ids": T(ids)[None].to(dev),
          "weights": T(weight)[None].to(dev)}
    gt = {"sdf": T(sdf_t)[None].to(dev), "sdf_valid": T(valid)[None].to(dev), "sdf_signs": T(sign)[None].to(dev)}
    return net, mi, gt, rs, n, sdf_t


def test_miso_losses_match_reference(device_backend):
    import miso_amd.grid_opt.loss as L
    import miso_amd.grid_opt.loss_isdf as LI
    dev = device_backend
    case = gc.CASES["small"]
    g = G("losses")
    net, mi, gt, rs, n, sdf_t = _loss_inputs(case, dev)
    np.testing.assert_array_equal(g["frame_ids"], mi["sample_frame_ids"][0].cpu().numpy())
    for lt in ("L1", "L2"):
        lossf = L.MisoLossMapping(loss_type=lt, weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
        net.zero_grad(set_to_none=True)
        d = lossf.compute(net, mi, gt)
        # column 0 of the model output is a slice: the reference's out[:, [0]] differentiates into a sort-based
        # index_put over all rows (1.6 ms at 540 000 rows)
        xw = lossf.world_coords(net, mi["coords_frame"][0], mi["sample_frame_ids"][0, :, 0])
        assert "Index" not in type(lossf.query_model(net, xw)["sdf"].grad_fn).__name__
        sum(v.mean() for v in d.values()).backward()
        assert set(d) == {f"sdf_{lt}", "free_space"}
        for k_, v in d.items():
            assert abs(v.item() - float(g[f"map_{lt}_{k_}"])) < 2e-6, k_
        close(net.rotation_corrections.grad, T(g[f"map_{lt}_gdr"]), 2e-4)
        close(net.translation_corrections.grad, T(g[f"map_{lt}_gdt"]), 2e-4)
        for l in range(case["n_levels"]):
            close(net.features[l].feature.grad, T(g[f"map_{lt}_gfeat{l}"]), 1e-4)
    gt_all = dict(gt)
    gt_all["sdf_valid"] = torch.ones_like(gt["sdf_valid"])
    for lt in ("L1", "L2", "GM"):
        lossf = L.MisoLossTracking(weight_sdf=1.0, loss_type=lt, trunc_dist=0.12, gm_scale_sdf=0.1)
        net.zero_grad(set_to_none=True)
        d = lossf.compute(net, mi, gt_all)
        sum(v.mean() for v in d.values()).backward()
        for k_, v in d.items():
            assert abs(v.item() - float(g[f"track_{lt}_{k_}"])) < 2e-6
        close(net.rotation_corrections.grad, T(g[f"track_{lt}_gdr"]), 2e-4)
        close(net.translation_corrections.grad, T(g[f"track_{lt}_gdt"]), 2e-4)
    # iSDF helpers (pure elementwise) + compute_slam through the model
    sdf_p, bounds = T(g["isdf_sdf"]).to(dev), T(g["isdf_bounds"]).to(dev)
    for lt in ("L1", "L2"):
        mat, fsix = LI.sdf_loss(sdf_p.clone(), bounds, 0.15, loss_type=lt)
        tot, tot_mat, _ = LI.tot_loss(mat, None, None, fsix, bounds, 0.1, 5.0, 0.0, 0.0)
        assert abs(tot.item() - float(g[f"isdf_{lt}_total"])) < 1e-6
        close(tot_mat, T(g[f"isdf_{lt}_mat"]), 1e-6, 1e-7)
    il = LI.iSDFLoss("grid_net", trunc_weight=5.0, trunc_distance=0.15, loss_type="L1", slam_mode=True)
    net.zero_grad(set_to_none=True)
    d = il.compute(net, mi, {"sdf": T(np.abs(sdf_t))[None].to(dev)})
    d["sdf"].backward()
    assert abs(d["sdf"].item() - float(g["isdf_slam_sdf"])) < 2e-6
    close(net.rotation_corrections.grad, T(g["isdf_slam_gdr"]), 2e-4)
    for l in range(case["n_levels"]):
        close(net.features[l].feature.grad, T(g[f"isdf_slam_gfeat{l}"]), 1e-4)


@pytest.mark.parametrize("captured", [True, False])
@pytest.mark.parametrize("mode", ["joint", "coordinate+joint"])
def test_grid_trainer_matches_reference(device_backend, mode, captured, tmp_path):
    """GridTrainer, 6 epochs, level switch every 2 epochs -> features equal the reference's
    (dense Adam semantics incl. untouched stability grids).  captured: on the GPU the step runs
    as one graph replay (Trainer._captured_mapping_step); False forces the op-by-op autograd path."""
    import miso_amd.grid_opt.loss as L
    from miso_amd.grid_opt.trainer import GridTrainer
    dev = device_backend
    case = gc.CASES["small"]
    g = G("trainer")
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t, valid, sign, weight = gc.make_targets(case, n)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return ({"coords_frame": T(pts), "sample_frame_ids": torch.zeros(n, 1, dtype=torch.int64),
                     "weights": T(weight)},
                    {"sdf": T(sdf_t), "sdf_valid": T(valid), "sdf_signs": T(sign)})

    net = make_gridnet(case, dev, stability=True)
    net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
    net.unlock_feature()
    net.lock_pose()
    cfg_train = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 6, "ckpt_every": -1,
                 "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": str(tmp_path),
                 "relchange_tol": 0, "max_epochs_in_level": 2, "grid_training_mode": mode,
                 "captured_step": captured}
    lossf = L.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
    loader = torch.utils.data.DataLoader(DS(), batch_size=1, shuffle=False, num_workers=0)
    tr = GridTrainer(cfg_train, net, lossf, loader, None, dev, torch.float32)
    tr.train()
    assert bool(tr.__dict__.get("_mapping_steps")) == (captured and str(dev).startswith("cuda"))
    tag = mode.replace("+", "_")
    for l in range(case["n_levels"]):
        close(net.features[l].feature, T(g[f"{tag}_feat{l}"]), 0, 3e-6)
        close(net.feature_stability[l].feature, T(g[f"{tag}_stab{l}"]), 0, 0)


def make_atlas(dev):
    from miso_amd.grid_opt.models.grid_atlas import GridAtlas
    c = gc.ATLAS
    cfg = gc.model_cfg(c["bound"], c["base_cell"], c["scale"], c["n_levels"], c["fdim"], c["hidden"])
    atlas = GridAtlas(cfg, device=dev)
    dec = {k: T(v) for k, v in gc.make_decoder(c).items()}
    for s, sub in enumerate(gc.atlas_inputs()):
        atlas.add_submap(torch.tensor(c["bound"], dtype=torch.float32), T(sub["R"]), T(sub["t"]), num_poses=2)
        atlas.add_kf(torch.eye(3), torch.zeros(3, 1))
        net = atlas.get_submap(s)
        with torch.no_grad():
            for l, f in enumerate(sub["features"]):
                net.features[l].feature.copy_(T(f))
        net.decoder.load_state_dict(dec)
        atlas.set_submap_pose_correction(s, T(sub["dr"]).to(dev), T(sub["dt"]).to(dev))
    return atlas.to(dev)


def test_atlas_queries_and_pairwise_alignment(device_backend):
    import miso_amd.grid_opt.align.base as AB
    import miso_amd.grid_opt.align.miso as AM
    dev = device_backend
    c = gc.ATLAS
    g = G("atlas")
    atlas = make_atlas(dev)
    xw = T(gc.atlas_world_points()).to(dev)
    close(atlas.query_feature(xw), T(g["query_feature"]), 0, 2e-6)
    close(atlas(xw), T(g["forward"]), 0, 1e-5)
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    for s in range(c["n_submaps"]):
        for l in range(c["n_levels"]):
            assert atlas.coordinates_for_alignment(s, l).shape[0] == int(g[f"ncoords_s{s}_l{l}"])
    for (a, b) in [(0, 1), (0, 2), (1, 2)]:
        assert bool(atlas.check_submap_intersection(a, b)) == bool(g[f"intersect_{a}_{b}"])
        for l in range(c["n_levels"]):
            for lt in ("L2", "L1"):
                atlas.zero_grad(set_to_none=True)
                d = AM.pairwise_loss_latent(atlas, None, a, b, level=l, fdim=c["fdim"], align_loss=lt, device=dev)
                (val,) = d.values()
                key = f"latent_{a}_{b}_l{l}_{lt}"
                assert list(d) == [f"align_latent_level{l}_{a}_{b}"]
                assert abs(val.item() - float(g[key])) <= 3e-5 * abs(float(g[key])), key
                val.backward()
                for which, s in (("src", a), ("dst", b)):
                    close(atlas.rotation_corrections[s].grad, T(g[key + f"_gR_{which}"]), 2e-3, 2e-3)
                    close(atlas.translation_corrections[s].grad, T(g[key + f"_gt_{which}"]), 2e-3, 2e-3)
                # features are read-only here: no dense grad is produced for them
                assert all(gr.feature.grad is None for sm in atlas.submaps for gr in sm.features)
    # multi-submap Adam alignment, 3 (+1) iterations per level, pose trajectory
    atlas.zero_grad(set_to_none=True)
    for l in range(c["n_levels"]):
        tup = (f"latent{l}", lambda at, ld, a, b, _l=l: AM.pairwise_loss_latent(
            at, ld, a, b, level=_l, fdim=c["fdim"], align_loss="L2", device=dev))

        class DS(torch.utils.data.Dataset):
            def __len__(self):
                return 1

            def __getitem__(self, i):
                return 0

        info = AB.generic_align_multiple_submaps(atlas, DS(), tup, num_iters=3, lr=1e-2, verbose=False,
                                                 save_iterations=(l == 0))
        assert set(info) == {"cpu_time_sec", "gpu_time_sec", "iteration_results"}
        if l == 0:
            assert sorted(info["iteration_results"]) == [0, 1, 2, 3] and info["iteration_results"][0].shape == (3, 4, 4)
        dr = torch.stack([p.detach().cpu() for p in atlas.rotation_corrections])
        dt = torch.stack([p.detach().cpu() for p in atlas.translation_corrections])
        close(dr, T(g[f"align_l{l}_dr"]), 0, 2e-4)
        close(dt, T(g[f"align_l{l}_dt"]), 0, 2e-4)


def test_pairwise_loss_latent_optional_branches(device_backend):
    """The branches of pairwise_loss_latent the default alignment never takes (reference align/miso.py:147-180,
    :204-209): stability pruning, truncation pruning, np.random.choice subsampling (the reference's own draw, recorded),
    the cos and InfoNCE losses -- value and the pose gradients of both submaps against the reference run
    (tests/golden/atlas_branches.npz, tools/make_goldens.py::gen_atlas_branches)."""
    import miso_amd.grid_opt.align.miso as AM
    dev = device_backend
    c, b = gc.ATLAS, gc.ATLAS_BRANCHES
    g = G("atlas_branches")
    atlas = make_atlas(dev)
    for s in range(c["n_submaps"]):
        net = atlas.get_submap(s)
        with torch.no_grad():
            for l, f in enumerate(gc.atlas_stability(s)):
                net.feature_stability[l].feature.copy_(T(f))
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    src, dst = b["pair"]
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
        atlas.zero_grad(set_to_none=True)
        if "subsample_points" in kw:
            np.random.seed(b["subsample_seed"])
            n_all = atlas.coordinates_for_alignment(src, kw["level"]).shape[0]
            draw = np.random.choice(n_all, min(kw["subsample_points"], n_all), replace=False)
            assert np.array_equal(draw, g[name + "_draw"])              # the reference drew exactly these vertices
            np.random.seed(b["subsample_seed"])
        d = AM.pairwise_loss_latent(atlas, None, src, dst, fdim=c["fdim"], device=dev, **kw)
        (val,) = d.values()
        assert list(d) == [f"align_latent_level{kw['level']}_{src}_{dst}"]
        assert abs(val.item() - float(g[name])) <= 5e-5 * abs(float(g[name])), (name, val.item(), float(g[name]))
        val.backward()
        for which, s in (("src", src), ("dst", dst)):
            gR, gt = T(g[f"{name}_gR_{which}"]), T(g[f"{name}_gt_{which}"])
            tol = 2e-3 * max(float(gR.abs().max()), float(gt.abs().max()))
            close(atlas.rotation_corrections[s].grad, gR, 0, tol)
            close(atlas.translation_corrections[s].grad, gt, 0, tol)


class _OneItem(torch.utils.data.Dataset):
    def __len__(self):
        return 1

    def __getitem__(self, i):
        return 0


def test_hierarchical_alignment_fused_loop_matches_reference(device_backend, monkeypatch):
    """align_multiple_submaps_hierarchical takes the fused loop (ops.AlignPlan: every pair of an iteration in one
    launch, overlap gate / NaN guard / Adam on the device; the oracle stand-in on CPU) and reproduces the
    reference's pose trajectory (3 (+1) Adam iterations per level, levels in sequence) -- with the reference's own
    alignment settings verbose=True, save_iterations=True (configs/rgbd/scannet.yaml:65-66), whose per-iteration
    (S,4,4) snapshots must equal those of the op-by-op loop."""
    from miso_amd import ops
    import miso_amd.grid_opt.align.miso as AM
    dev = device_backend
    c = gc.ATLAS
    g = G("atlas")
    atlas = make_atlas(dev)
    calls = []
    real = ops.AlignPlan.iteration_a
    monkeypatch.setattr(ops.AlignPlan, "iteration_a", lambda self: (calls.append(self.P), real(self))[1])
    info = AM.align_multiple_submaps_hierarchical(atlas, _OneItem(), level_iters=3, lr=1e-2, align_loss="L2",
                                                  skip_finetune=True, device=dev, verbose=True, save_iterations=True)
    assert calls == [3] * (c["n_levels"] * 4)         # num_iters + 1 iterations per level, all 3 pairs in each
    assert {f"hier_latent_level{l}_L2" for l in range(c["n_levels"])} <= set(info)
    l = c["n_levels"] - 1
    dr = torch.stack([p.detach().cpu() for p in atlas.rotation_corrections])
    dt = torch.stack([p.detach().cpu() for p in atlas.translation_corrections])
    close(dr, T(g[f"align_l{l}_dr"]), 0, 2e-4)
    close(dt, T(g[f"align_l{l}_dt"]), 0, 2e-4)
    # the same run op by op (autograd through so3_exp_map, per-pair losses, DenseAdam): snapshots agree
    ref = make_atlas(dev)
    ref.no_fused_alignment = True
    info_ref = AM.align_multiple_submaps_hierarchical(ref, _OneItem(), level_iters=3, lr=1e-2, align_loss="L2",
                                                      skip_finetune=True, device=dev, verbose=False, save_iterations=True)
    for l in range(c["n_levels"]):
        a, b = info[f"hier_latent_level{l}_L2"]["iteration_results"], info_ref[f"hier_latent_level{l}_L2"]["iteration_results"]
        assert sorted(a) == sorted(b) == [0, 1, 2, 3]
        for it in a:
            assert a[it].shape == (c["n_submaps"], 4, 4)
            close(a[it].cpu(), b[it].cpu(), 0, 2e-4)


def test_fused_alignment_regulariser_nan_guard_and_early_stop(device_backend):
    """The branches of generic_align_multiple_submaps around the pair losses, through the fused loop: the
    trust-region regulariser (base.py:20-27,143-145) equals the op-by-op loop; a NaN loss skips the step
    (base.py:147-151) and leaves the poses alone; rel_change_thresh stops the loop (base.py:157-158) at the same
    iteration as the op-by-op loop."""
    import miso_amd.grid_opt.align.base as AB
    import miso_amd.grid_opt.align.miso as AM
    dev = device_backend
    c = gc.ATLAS

    def run(fused, **kw):
        atlas = make_atlas(dev)
        atlas.no_fused_alignment = not fused
        atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
        loss = AM.latent_loss_for_level(atlas, 0, align_loss="L1", device=dev)     # level 0: seconds on the CPU stand-in
        assert hasattr(loss, "fused")
        info = AB.generic_align_multiple_submaps(atlas, _OneItem(), ("latent1", loss), lr=1e-2, verbose=False, **kw)
        dr = torch.stack([p.detach().cpu() for p in atlas.rotation_corrections])
        dt = torch.stack([p.detach().cpu() for p in atlas.translation_corrections])
        return dr, dt, info, atlas

    kw = dict(num_iters=4, pose_reg_weight=1.0, pose_thresh_rad=1e-3, pose_thresh_m=1e-3)
    dr_f, dt_f, _, _ = run(True, **kw)
    dr_e, dt_e, _, _ = run(False, **kw)
    close(dr_f, dr_e, 0, 2e-4)
    close(dt_f, dt_e, 0, 2e-4)
    # early stop: both loops must break after the same iteration
    # (relative changes on this problem: 0.122 at iteration 4, 0.109 at iteration 5)
    kw = dict(num_iters=10, rel_change_thresh=0.115, save_iterations=True)
    dr_f, dt_f, info_f, _ = run(True, **kw)
    dr_e, dt_e, info_e, _ = run(False, **kw)
    assert sorted(info_e["iteration_results"]) == [0, 1, 2, 3, 4, 5]
    assert sorted(info_f["iteration_results"]) == sorted(info_e["iteration_results"])
    close(dr_f, dr_e, 0, 3e-4)
    close(dt_f, dt_e, 0, 3e-4)
    # NaN: a NaN base translation poisons every pair (nan_to_num'ed to 0, no gradient) and a NaN correction the
    # regulariser -> total loss NaN -> no step is taken
    atlas = make_atlas(dev)
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    with torch.no_grad():
        atlas.translation_corrections[2][0, 0] = float("nan")
    before = [p.detach().clone() for p in atlas.params_for_all_submap_poses()]
    loss = AM.latent_loss_for_level(atlas, 0, device=dev)
    AB.generic_align_multiple_submaps(atlas, _OneItem(), ("latent0", loss), num_iters=3, lr=1e-2, verbose=False,
                                      pose_reg_weight=1.0)
    for p, q in zip(atlas.params_for_all_submap_poses(), before):
        assert torch.equal(torch.nan_to_num(p.detach(), nan=7.0), torch.nan_to_num(q, nan=7.0))


def test_fused_alignment_leaves_submaps_without_a_gradient_alone(device_backend):
    """torch.optim.Adam skips a parameter whose .grad is None -- in generic_align_multiple_submaps a submap
    none of whose pairs is in the loss (skipped by check_submap_intersection, or not in submap_pairs) with no
    regulariser.  Neither its value nor its moments nor its step count move; the other submaps' bias corrections run
    on their OWN step counts.  (The fused loop used to step all 6(S-1) numbers with one global count: such a submap
    kept drifting on decaying momentum.)"""
    from miso_amd import ops
    import miso_amd.grid_opt.align.base as AB
    import miso_amd.grid_opt.align.miso as AM
    dev = device_backend

    def run(fused):
        atlas = make_atlas(dev)
        atlas.no_fused_alignment = not fused
        atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
        loss = AM.latent_loss_for_level(atlas, 0, align_loss="L2", device=dev)
        AB.generic_align_multiple_submaps(atlas, _OneItem(), ("latent0", loss), num_iters=4, lr=1e-2, verbose=False,
                                          submap_pairs=[(0, 1)])
        return (torch.stack([p.detach().cpu() for p in atlas.rotation_corrections]),
                torch.stack([p.detach().cpu() for p in atlas.translation_corrections]))

    start = make_atlas(dev)
    dr_f, dt_f = run(True)
    dr_e, dt_e = run(False)
    close(dr_f, dr_e, 0, 2e-4)
    close(dt_f, dt_e, 0, 2e-4)
    assert torch.equal(dr_f[2], start.rotation_corrections[2].detach().cpu())       # submap 2: in no pair
    assert torch.equal(dt_f[2], start.translation_corrections[2].detach().cpu())
    assert not torch.equal(dr_f[1], start.rotation_corrections[1].detach().cpu())
    # ... also when it carries momentum from an earlier iteration in which it did have a gradient (a gate that closed)
    atlas = make_atlas(dev)
    atlas.precompute_coordinates_for_alignment(norm_thresh=1e-5)
    S = atlas.num_submaps
    R0 = torch.stack([R.to(dev) for R in atlas.R_world_submap_list])
    t0 = torch.stack([t.to(dev) for t in atlas.t_world_submap_list])
    plan = ops.AlignPlan(R0, t0, AM.latent_pair_inputs(atlas, [(0, 1)], level=0, fdim=gc.ATLAS["fdim"]), lr=1e-2,
                         ring_iters=4)
    prm0 = torch.cat((torch.cat([p.detach().reshape(1, 3) for p in atlas.rotation_corrections]),
                      torch.cat([p.detach().reshape(1, 3) for p in atlas.translation_corrections])), 1)
    plan.params.copy_(prm0)
    mom = torch.full((6,), 0.3)
    if hasattr(plan, "state"):                                   # HIP: exp_avg lives in the device state
        plan._view(6, 6 * S).view(S, 6)[2].copy_(mom)
    else:
        plan.m[2] = mom.clone()
    for _ in range(3):
        plan.iteration_a()
        plan.iteration_b()
    prm = plan.params.detach().cpu()
    assert torch.equal(prm[2], prm0[2].cpu()) and not torch.equal(prm[1], prm0[1].cpu())
    if hasattr(plan, "state"):
        assert torch.equal(plan._view(6, 6 * S).view(S, 6)[2].cpu(), mom)
        assert plan.adam_steps.cpu().tolist() == [0, 3, 0]
    else:
        assert torch.equal(plan.m[2], mom) and plan.steps == [0, 3, 0]


@pytest.mark.parametrize("lt", ["GM", "L2"])
def test_tracker_lm_step_matches_reference(device_backend, lt):
    from miso_amd.grid_opt.slam.tracker import Tracker
    dev = device_backend
    case = gc.CASES["small"]
    g = G("tracker")
    net = make_gridnet(case, dev, num_poses=2, optimize_pose=True)
    net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
    net.set_initial_kf_pose(1, T(g["R0"]), T(g["t0"]), kf_key="KF1")
    pts = gc.make_points(case)
    n = pts.shape[0]

    class DS(torch.utils.data.Dataset):
        def select_keyframes(self, kfs):
            pass

        def __len__(self):
            return 1

        def __getitem__(self, i):
            return ({"coords_frame": T(pts), "sample_frame_ids": torch.ones(n, 1, dtype=torch.int64),
                     "weights": torch.ones(n, 1)},
                    {"sdf": T(g["sdf"]), "sdf_valid": torch.ones(n, 1), "sdf_signs": torch.zeros(n, 1)})

    cfg = {"device": dev, "train": {},
           "tracking": {"learning_rate": 1e-3, "verbose": False, "gm_scale_sdf": 0.1, "lm_lambda": 5.0,
                        "lm_max_iter": 3, "lm_tol_deg": 0.0, "lm_tol_m": 0.0, "loss_type": lt,
                        "trunc_dist": 0.12, "solver": "lm"}}
    trk = Tracker(net, DS(), cfg)
    for it in range(3):
        info = trk.lm_step(1)
        ref = g[f"lm_{lt}_info"][it]
        got = [info["delta_R_deg"], info["delta_t_norm"], info["grad_norm"], info["fov_overlap"]]
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-6)
        close(net.rotation_corrections, T(g[f"lm_{lt}_dr_{it}"]), 2e-3, 1e-6)
        close(net.translation_corrections, T(g[f"lm_{lt}_dt_{it}"]), 2e-3, 1e-6)


def test_compat_aliases_and_pickle_roundtrip(device_backend, tmp_path):
    """`import grid_opt...` / `cuda_gridsample` resolve to this package, and a whole-module
    pickle (demo/build_submaps.py:141 -> demo/align_submaps.py:263) round-trips."""
    import miso_amd.compat  # noqa: F401
    import cuda_gridsample
    import grid_opt.models.grid_net as alias
    import miso_amd.grid_opt.models.grid_net as real
    assert alias is real and callable(cuda_gridsample.grid_sample_3d)
    dev = device_backend
    atlas = make_atlas(dev)
    path = tmp_path / "grid_atlas.pth"
    torch.save(atlas, path)
    back = torch.load(path, weights_only=False)
    xw = T(gc.atlas_world_points()).to(dev)
    close(back(xw), atlas(xw), 0, 0)
    assert back.get_submap(1).features[1].feature.is_contiguous(memory_format=torch.channels_last_3d)


def test_extract_fields_matches_pointwise_queries(device_backend):
    """Slab-wise dense extraction == the reference's per-lattice-point values (utils_sdf.py:69-86)."""
    from miso_amd.grid_opt.utils.utils_sdf import extract_fields
    dev = device_backend
    case = gc.CASES["small"]
    net = make_gridnet(case, dev)
    b = torch.tensor(case["bound"])
    res = 21
    u = extract_fields(b[:, 0], b[:, 1], res, lambda p: net(p), device=dev, max_points=res * res * 4)
    assert u.shape == (res, res, res) and u.dtype == np.float32
    ax = [torch.linspace(float(b[a, 0]), float(b[a, 1]), res) for a in range(3)]
    xx, yy, zz = torch.meshgrid(*ax, indexing="ij")
    pts = torch.stack((xx, yy, zz), -1).reshape(-1, 3)
    from oracle import ref_torch as R
    ws, bs = R.decoder_params({k: T(v) for k, v in gc.make_decoder(case).items()})
    ref = R.sdf_stock([T(f) for f in gc.make_features(case)], b, pts, ws, bs).reshape(res, res, res)
    close(T(u), ref, 0, 1e-5)


# ---- rows pinned by tests/golden/extra.npz (tools/make_goldens.py gen_extra) ---------------------------------------
class _OneBatch(torch.utils.data.Dataset):
    def __init__(self, mi, g):
        self.mi, self.g = mi, g

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return ({k: T(v[0]) for k, v in self.mi.items()}, {k: T(v[0]) for k, v in self.g.items()})


def make_atlas_two_kf(dev):
    """make_atlas with a second, non-identity keyframe in every submap (global keyframe ids 2s, 2s+1)."""
    from miso_amd.grid_opt.models.grid_atlas import GridAtlas
    c = gc.ATLAS
    cfg = gc.model_cfg(c["bound"], c["base_cell"], c["scale"], c["n_levels"], c["fdim"], c["hidden"])
    atlas = GridAtlas(cfg, device=dev)
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
        atlas.set_submap_pose_correction(s, T(sub["dr"]).to(dev), T(sub["dt"]).to(dev))
    return atlas.to(dev)


def test_grid_pool_3d_avg_matches_reference(device_backend):
    """Scatter-average pooling of point features onto a regular grid (reference utils.py:239-291): crowded cells,
    points outside the bound clamped into the border cells, empty cells stay zero."""
    from miso_amd.grid_opt.utils.utils import grid_pool_3d_avg
    dev = device_backend
    pts, feats = gc.pool_inputs()
    out = grid_pool_3d_avg(T(pts).to(dev), T(feats).to(dev), torch.tensor(gc.POOL["bound"], dtype=torch.float32).to(dev),
                           gc.POOL["cell"])
    close(out, T(G("extra")["pool"]), 0, 2e-6)


def test_pairwise_loss_sdf_matches_reference(device_backend):
    """SDF-space pair loss (reference align/miso.py:14-113): observed rows of src's keyframes through the per-keyframe
    rigid maps, src -> world -> dst, bound / validity masks, both submaps' encode + decode; L2 / L1 / GM values and the
    cotangents of both submaps' pose corrections."""
    import miso_amd.grid_opt.align.miso as AM
    dev = device_backend
    g = G("extra")
    atlas = make_atlas_two_kf(dev)
    mi, gt = gc.atlas_sdf_batch()
    loader = torch.utils.data.DataLoader(_OneBatch(mi, gt), batch_size=1, shuffle=False, num_workers=0)
    for (a, b) in [(0, 1), (1, 2), (2, 0)]:
        for lt in ("L2", "L1", "GM"):
            atlas.zero_grad(set_to_none=True)
            d = AM.pairwise_loss_sdf(atlas, loader, a, b, align_loss=lt, device=dev)
            assert list(d) == [f"align_sdf_{a}_{b}"]
            (val,) = d.values()
            key = f"pairsdf_{a}_{b}_{lt}"
            assert abs(val.item() - float(g[key])) <= 3e-5 * abs(float(g[key])), (key, val.item(), float(g[key]))
            val.backward()
            for which, s in (("src", a), ("dst", b)):
                close(atlas.rotation_corrections[s].grad, T(g[key + f"_gR_{which}"]), 2e-3, 2e-3)
                close(atlas.translation_corrections[s].grad, T(g[key + f"_gt_{which}"]), 2e-3, 2e-3)


def test_local_opt_matches_reference(device_backend, tmp_path):
    """local_opt.optimize_grid_net (iSDF loss) and optimize_grid_atlas (iSDFSubmap loss) through GridTrainer
    (reference local_opt.py:60-154): features after 5 joint / 4 coordinate iterations, info dict contract."""
    import miso_amd.grid_opt.local_opt as LO
    dev = device_backend
    g = G("extra")
    case = gc.CASES["small"]
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t = gc.make_targets(case, n)[0]
    cfg = gc.local_opt_cfg("iSDF")
    cfg["device"] = dev
    cfg["train"]["log_dir"] = str(tmp_path)
    net = make_gridnet(case, dev, stability=True)
    net.unlock_feature()
    net.lock_pose()
    ds = _OneBatch({"coords": pts[None], "normals": np.zeros((1, n, 3), np.float32)},
                   {"sdf": sdf_t[None], "grad_vec": np.zeros((1, n, 3), np.float32)})
    net2, info = LO.optimize_grid_net(net, ds, cfg, iterations=5, learning_rate=2e-3, train_mode="joint",
                                      iterations_per_level=2)
    assert net2 is net and set(info) == {"trainer_epoch", "trainer_epoch_time", "trainer_total_loss"}
    for l in range(case["n_levels"]):
        close(net.features[l].feature, T(g[f"localopt_net_feat{l}"]), 0, 3e-6)

    atlas = make_atlas_two_kf(dev)
    mi, gt = gc.atlas_sdf_batch()
    owner = (mi["sample_frame_ids"] // 2).astype(np.int64)
    ds = _OneBatch({"coords_submap": mi["coords_frame"], "submap_idxs": owner}, {"sdf": gt["sdf"], "sdf_valid": gt["sdf_valid"]})
    cfg = gc.local_opt_cfg("iSDFSubmap")
    cfg["device"] = dev
    cfg["train"]["log_dir"] = str(tmp_path)
    atlas2, info = LO.optimize_grid_atlas(atlas, ds, cfg, iterations=4, learning_rate=2e-3, train_mode="coordinate")
    assert atlas2 is atlas and info == {}
    for s in range(gc.ATLAS["n_submaps"]):
        for l in range(gc.ATLAS["n_levels"]):
            close(atlas.get_submap(s).features[l].feature, T(g[f"localopt_atlas_s{s}_feat{l}"]), 0, 3e-6)
    close(torch.stack([p for p in atlas.rotation_corrections]), T(g["localopt_atlas_dr"]), 0, 3e-6)
    close(torch.stack([p for p in atlas.translation_corrections]), T(g["localopt_atlas_dt"]), 0, 3e-6)


def test_geometry_helpers_match_reference(tmp_path):
    """utils_geometry beyond the rigid maps (frame changes of pose sets, batched world transform with the anchored
    first frame, AABB, voxel down-sampling, range crop, error metrics, random pose perturbations from numpy's global
    RNG, KITTI pose files) against rows produced by the reference (tests/golden/geometry.npz)."""
    import miso_amd.grid_opt.utils.utils_geometry as UG
    g = G("geometry")
    x = gc.geometry_inputs()
    close(UG.batch_transform_to_world_frame(T(x["pts"]), T(x["spans"]), T(x["R"]), T(x["t"]), T(x["dr"]), T(x["dt"])),
          T(g["batch_world"]), 0, 3e-6)
    Rf, tf = UG.transform_poses_from(T(x["R"]), T(x["t"]), T(x["R"][1]), T(x["t"][1]))
    close(Rf, T(g["poses_from_R"]), 0, 1e-6)
    close(tf, T(g["poses_from_t"]), 0, 2e-6)
    close(UG.aabb_torch(T(x["cloud"]), buffer=0.25), T(g["aabb"]), 0, 0)
    for vs in (0.5, 2.0):
        assert np.array_equal(UG.voxel_down_sample_torch(T(x["cloud"]), vs).numpy(), g[f"voxel_{vs}"])
    p, s = UG.crop_points(T(x["cloud"]), T(x["stamps"]), min_z_th=-1.0, max_z_th=2.0, min_range=2.75, max_range=9.0)
    assert np.array_equal(p.numpy(), g["crop_pts"]) and np.array_equal(s.numpy(), g["crop_ts"])
    assert UG.crop_points(T(x["cloud"]), None)[1] is None
    assert abs(UG.translation_rmse(T(x["t"]), T(x["dt"])) - float(g["t_rmse"])) < 1e-6
    assert abs(UG.translation_mean_error(T(x["t"]), T(x["dt"])) - float(g["t_mean"])) < 1e-6
    assert abs(UG.chordal_to_degree(0.7) - float(g["chordal_deg"])) < 1e-12
    np.random.seed(7)
    close(UG.gaussian_translations(5, 0.5), T(g["gauss_t"]), 0, 0)
    close(UG.uniform_translations(5, np.array([[-1.0, 1.0], [0.0, 2.0], [3.0, 4.0]])), T(g["uniform_t"]), 0, 0)
    close(UG.fixed_length_translations(5, 0.3), T(g["fixed_len_t"]), 0, 1e-7)
    close(UG.wrapped_gaussian_rotations(5, std_rad=0.2), T(g["wrapped_R"]), 0, 1e-6)
    close(UG.fixed_angle_rotations(5, 0.4), T(g["fixed_angle_R"]), 0, 1e-6)
    poses = np.tile(np.eye(4), (4, 1, 1))
    poses[:, :3, :3], poses[:, :3, 3:] = x["R"], x["t"]
    UG.write_kitti_format_poses(str(tmp_path / "traj"), poses)
    assert open(tmp_path / "traj_kitti.txt", "rb").read() == g["kitti_text"].tobytes()
    assert np.array_equal(np.stack(UG.read_kitti_format_poses(str(tmp_path / "traj_kitti.txt"))), g["kitti_read"])
    (tmp_path / "bad.txt").write_text("1 2 3\n")
    assert UG.read_kitti_format_poses(str(tmp_path / "bad.txt")) is None
    ok = [UG.check_numpy_pose_matrix(poses[0]), UG.check_numpy_pose_matrix(poses[0] * 1.01),
          UG.check_numpy_pose_matrix(np.full((4, 4), np.nan))]
    assert ok == [bool(v) for v in g["pose_ok"]]
    # the two helpers whose pytorch3d calls are not shimmed: pinned by their defining properties
    R0, t0 = T(x["R"][0]), T(x["t"][0])
    for k in range(1, 4):
        dr, dt = UG.get_pose_correction(R0, t0, T(x["R"][k]), T(x["t"][k]))
        assert dr.shape == (1, 3) and dt.shape == (3, 1)
        Rn, tn = UG.apply_pose_correction(R0, t0, dr, dt)
        close(Rn, T(x["R"][k]), 0, 3e-6)
        close(tn, T(x["t"][k]), 0, 1e-6)
    close(UG.get_pose_correction(R0, t0, R0, t0)[0], torch.zeros(1, 3), 0, 1e-6)
    ang = torch.tensor([[0.0, 0.0, 0.1], [0.3, 0.0, 0.0], [0.0, -0.2, 0.0]])
    Ra = UG.so3_exp_map(ang)
    eye = UG.identity_rotations(3)
    assert abs(UG.rotation_rmse(Ra, eye) - np.degrees(np.sqrt((0.01 + 0.09 + 0.04) / 3))) < 1e-3
    assert abs(UG.rotation_mean_error(Ra, eye) - np.degrees(0.2)) < 1e-3


def test_files_written_by_the_reference_load(device_backend, tmp_path):
    """a Trainer.save_model checkpoint (reference trainer.py:319-332) and a whole-module pickle
    torch.save(grid_atlas) (demo/build_submaps.py:141), both written by the reference itself
    (tools/make_goldens.py gen_formats), load through miso_amd.compat: same state-dict keys, same dotted class
    paths, and the loaded objects answer queries like the reference did before saving."""
    import miso_amd.compat  # noqa: F401
    import miso_amd.grid_opt.loss as L
    from miso_amd.grid_opt.models.grid_atlas import GridAtlas
    from miso_amd.grid_opt.models.grid_net import GridNet
    from miso_amd.grid_opt.trainer import GridTrainer
    dev = device_backend
    g = G("formats")
    case = gc.CASES["small"]
    ck_path = os.path.join(gc.GOLDEN_DIR, "ref_checkpoint.pt")
    ck = torch.load(ck_path, weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "train_dict", "val_dict"}
    net = make_gridnet(case, dev)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["ckpt_keys"]]
    # through the trainer's own loading path (cfg['pretrained_model'])
    with torch.no_grad():
        for f in net.features:
            f.feature.zero_()
    cfg_train = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 0, "ckpt_every": -1,
                 "eval_every": -1, "eval_metric": None, "pretrained_model": ck_path, "log_dir": str(tmp_path),
                 "relchange_tol": 0, "max_epochs_in_level": 2, "grid_training_mode": "joint"}
    lossf = L.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
    GridTrainer(cfg_train, net, lossf, None, None, dev, torch.float32)
    xq = T(gc.make_points(case)[:256]).to(dev)
    close(net(xq), T(g["ckpt_forward"]), 0, 1e-5)
    # our own checkpoint of the loaded model carries the same keys
    assert set(net.state_dict().keys()) == set(ck["model_state_dict"].keys())

    atlas = torch.load(os.path.join(gc.GOLDEN_DIR, "ref_atlas.pth"), weights_only=False, map_location="cpu")
    assert type(atlas) is GridAtlas and type(atlas.get_submap(0)) is GridNet
    atlas.to(dev)
    xw = T(g["atlas_x"]).to(dev)
    close(atlas(xw), T(g["atlas_forward"]), 0, 1e-5)
    close(atlas.query_feature(xw), T(g["atlas_query_feature"]), 0, 2e-6)
    R, t = atlas.updated_kf_pose_in_world(3)
    close(R, T(g["atlas_kf3_R"]), 0, 1e-6)
    close(t, T(g["atlas_kf3_t"]), 0, 1e-6)
    assert [atlas.anchor_kf_for_submap(0), atlas.anchor_kf_for_submap(1)] == g["atlas_anchor"].tolist()
    # and it can be saved and loaded again
    torch.save(atlas, tmp_path / "again.pth")
    again = torch.load(tmp_path / "again.pth", weights_only=False)
    close(again(xw), T(g["atlas_forward"]), 0, 1e-5)


def test_encoder_initialisation_matches_reference(device_backend):
    """residual pooling -> FeaturePrediction (3-D convs, trilinear resampling, per-voxel MLP) ->
    per-level corrections, coarse to fine; the pre-training loss with its gradient to one level's predictor; and
    local_opt.initialize_grid_net(init_mode='encode') -- against the reference run with the same seeded predictor
    weights (tests/golden/encoder.npz; the weights travel as plain state-dict arrays)."""
    import miso_amd.grid_opt.local_opt as LO
    from miso_amd.grid_opt.models.encoder import Encoder, EncoderObservation, EncoderPretrainLoss
    dev = device_backend
    g = G("encoder")
    case = gc.CASES["small"]
    cfg = {"device": dev, "model": gc.model_cfg(case["bound"], case["base_cell"], case["scale"], case["n_levels"],
                                                case["fdim"], case["hidden"])}

    def make_encoder():
        enc = Encoder(cfg).to(dev)
        for l in range(2):
            sd = {k[len(f"enc{l}."):]: T(g[k]) for k in g.files if k.startswith(f"enc{l}.")}
            assert set(sd) == set(enc.feature_encoders[l].state_dict())          # upstream's keys
            enc.feature_encoders[l].load_state_dict(sd)
        return enc

    enc = make_encoder()
    assert not any(p.requires_grad for p in enc.feature_encoders.parameters())
    net = make_gridnet(case, dev)
    pts = gc.make_points(case)
    n = pts.shape[0]
    sdf_t, valid, sign, _ = gc.make_targets(case, n)
    obs = EncoderObservation(coords_world=T(pts).to(dev), gt_sdf=T(sdf_t).to(dev), gt_sdf_sign=T(sign).to(dev),
                             gt_sdf_valid=T(valid).to(dev))
    mid = enc.register_grid_model(net)
    assert mid == 0 and enc.correction_key(0, 1) == "gridnet0_correction_level1"
    res = enc.compute_residuals(mid, [torch.zeros_like(f.feature) for f in net.features], obs)
    for k in ("sdf_constraint", "fs_constraint", "fs_upper_constraint", "fs_lower_constraint"):
        close(res[k], T(g[f"res_{k}"]), 0, 2e-6)
    close(enc.compute_encoder_inputs_from_residuals(res, mid, 1), T(g["enc_inputs_l1"]), 0, 2e-6)
    corr = enc.predict_corrections_until_level(mid, 2, obs, pred_std=0, store_corrections=True)
    for l in range(2):
        close(corr[l], T(g[f"corr{l}"]), 0, 2e-5)
        close(enc.get_grid_correction(mid, l), T(g[f"corr{l}"]), 0, 2e-5)
    assert float(enc.stored_corrections_until_level(mid, 1)[1].abs().sum()) == float(g["stored_until1_l1_abs"])
    assert "encoder_inputs_model0_level1" in enc.intermediate_results

    enc.lock_all_params()
    enc.unlock_encoder_at_level(1)
    lossf = EncoderPretrainLoss(target_level=1, sdf_weight=3e3, sign_weight=10.0, pred_std=0.0)
    net2 = make_gridnet(case, dev, num_poses=2)
    mid2 = enc.register_grid_model(net2)
    spans = np.array([[0, n // 2], [n // 2, n]], dtype=np.int64)
    mi = {"dataset_index": torch.tensor([mid2]), "coords_frame": T(pts)[None].to(dev), "frame_indices": T(spans)[None],
          "R_world_frame": torch.eye(3).repeat(2, 1, 1)[None].to(dev), "t_world_frame": torch.zeros(1, 2, 3, 1, device=dev)}
    gt = {"sdf": T(sdf_t)[None].to(dev), "sdf_valid": T(valid)[None].to(dev), "sdf_signs": T(sign)[None].to(dev)}
    ld = lossf.compute(enc, mi, gt)
    assert set(ld) == {"sdf", "free_space"}
    assert abs(ld["sdf"].item() - float(g["pretrain_sdf"])) <= 2e-4 * float(g["pretrain_sdf"])
    assert abs(ld["free_space"].item() - float(g["pretrain_fs"])) <= 2e-4 * float(g["pretrain_fs"])
    sum(ld.values()).backward()
    for k, p in enc.feature_encoders[1].named_parameters():
        close(p.grad, T(g[f"pretrain_grad.{k}"]), 2e-3, 1e-4)
    assert all(p.grad is None for p in enc.feature_encoders[0].parameters())

    enc3 = make_encoder()
    net3 = make_gridnet(case, dev)
    net3b, info = LO.initialize_grid_net(net3, "encode", enc3, obs)
    assert net3b is net3 and set(info) == {"total_encoder_time"}
    for l in range(2):
        close(net3.features[l].feature, T(g[f"init_feat{l}"]), 0, 2e-5)
    # the other two modes
    LO.initialize_grid_net(net3, "zero")
    assert all(float(f.feature.detach().abs().sum()) == 0.0 for f in net3.features)
    LO.initialize_grid_net(net3, "randn")
    assert all(0 < float(f.feature.detach().std()) < 1e-3 for f in net3.features)


@pytest.mark.gpu
def test_captured_alignment_loop_equals_the_eager_one(caplog):
    """generic_align_multiple_submaps runs the pose-Adam loop on the device (fused iteration, replayed as one HIP
    graph from 8 iterations): same pose trajectory end point as the op-by-op loop, same info keys, and the
    per-iteration log lines the reference prints (losses and relative pose changes kept on the device, written out
    afterwards)."""
    import logging
    import miso_amd.grid_opt.align.miso as AM

    class _DS(torch.utils.data.Dataset):
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return 0

    def run(captured, verbose=False):
        atlas = make_atlas("cuda:0")
        atlas.no_fused_alignment = not captured
        for s in range(atlas.num_submaps):
            atlas.get_submap(s).lock_feature()
        info = AM.align_multiple_submaps_hierarchical(atlas, _DS(), level_iters=14, latent_levels=[0, 1], skip_finetune=True,
                                                      device="cuda:0", verbose=verbose)
        dr = torch.stack([p.detach().cpu() for p in atlas.rotation_corrections])
        dt = torch.stack([p.detach().cpu() for p in atlas.translation_corrections])
        return dr, dt, info

    dr_e, dt_e, info_e = run(False)
    dr_e2, dt_e2, _ = run(False)
    with caplog.at_level(logging.INFO, logger="miso_amd.grid_opt.align.base"):
        dr_c, dt_c, info_c = run(True, verbose=True)
    assert set(info_e) == set(info_c)
    # 30 Adam steps of 1e-2: Adam normalises the gradient, so rounding differences in near-zero components are
    # amplified.  Measured on this problem: eager vs eager 1e-5 (float atomics), the eager loop with DenseAdam vs
    # with torch.optim.Adam 4.4e-4, torch's capturable Adam eager vs captured 1.2e-4, eager vs captured 4.5e-4 --
    # the spread of "the same Adam" in fp32, not of the capture
    noise = max((dr_e - dr_e2).abs().max().item(), (dt_e - dt_e2).abs().max().item())
    tol = max(1.5e-3, 5 * noise)
    assert (dr_e - dr_c).abs().max().item() <= tol and (dt_e - dt_c).abs().max().item() <= tol, (noise, tol)
    first = gc.atlas_inputs()[0]
    close(dr_c[0], T(first["dr"]), 0, 0)                                       # submap 0 stays where it was
    close(dt_c[0], T(first["dt"]), 0, 0)
    lines = [r.getMessage() for r in caplog.records if "AlignMulti_hier_latent_level1_L2 iteration" in r.getMessage()]
    assert len(lines) == 15 and "pose_relchange=inf" in lines[0] and "iteration 14" in lines[-1]
    assert not any("not captured" in r.getMessage() for r in caplog.records)


@pytest.mark.gpu
def test_captured_mapping_step_ignores_stale_gradients_of_other_parameters(tmp_path):
    """Adam steps every parameter whose .grad is not None, whatever its requires_grad.  After an
    adam tracking window the keyframe pose corrections carry a gradient; the joint optimizer of the mapping phase
    holds them too (model.parameters()), and the captured step -- which bypasses optimizer.zero_grad() -- must not
    let that stale gradient move the locked poses (the reference's zero_grad(set_to_none=True) drops it)."""
    from miso_amd.grid_opt.loss import MisoLossMapping
    from miso_amd.grid_opt.models.grid_net import GridNet
    from miso_amd.grid_opt.trainer import GridTrainer
    dev = "cuda:0"
    c = gc.ATLAS
    cfg = gc.model_cfg(c["bound"], c["base_cell"], c["scale"], c["n_levels"], c["fdim"], c["hidden"], num_poses=2,
                       init_stddev=1e-2)
    torch.manual_seed(0)
    net = GridNet(cfg, device=dev).to(dev)
    for k in range(2):
        net.set_initial_kf_pose(k, torch.eye(3), torch.zeros(3, 1), kf_key=f"KF{k}")
    net.unlock_feature()
    net.lock_pose()
    n = 4096
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([1.8, 0.9, 1.8])
    mi = {"coords_frame": x[None].to(dev), "sample_frame_ids": torch.randint(0, 2, (1, n, 1), generator=g).to(dev),
          "weights": torch.ones(1, n, 1, device=dev)}
    gt = {"sdf": (torch.rand(1, n, 1, generator=g) * 0.2 - 0.1).to(dev), "sdf_valid": torch.ones(1, n, 1, device=dev),
          "sdf_signs": torch.zeros(1, n, 1, device=dev)}
    tcfg = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-2, "epochs": 1, "ckpt_every": -1, "eval_every": -1,
            "eval_metric": None, "pretrained_model": None, "log_dir": str(tmp_path), "relchange_tol": 0,
            "max_epochs_in_level": 1000, "grid_training_mode": "joint"}
    lf = MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
    tr = GridTrainer(tcfg, net, lf, None, None, dev, torch.float32)
    poses = list(net.params_for_poses())
    assert poses and not any(p.requires_grad for p in poses)
    before = [p.detach().clone() for p in poses]
    feat0 = net.features[0].feature.detach().clone()
    for _ in range(3):
        for p in poses:
            p.grad = torch.ones_like(p)          # what Tracker.track_window (solver 'adam') leaves behind
        tr.train_step(mi, gt)
    assert tr.__dict__.get("_mapping_steps"), "the captured step did not run"
    for p, q in zip(poses, before):
        assert torch.equal(p.detach(), q)
        assert p.grad is None
    assert not torch.equal(net.features[0].feature.detach(), feat0)      # the grids did train


def test_dense_adam_state_round_trips_with_torch_adam():
    """checkpoints are interchangeable -- a torch.optim.Adam state (the reference's
    optimizer_state_dict, trainer.py:319-329: 'step' is a float tensor) loads into DenseAdam and steps on; a DenseAdam
    state loads into torch.optim.Adam (needs weight_decay / amsgrad / maximize in the param group) and steps on; both
    continue the same trajectory."""
    from miso_amd.optim import DenseAdam
    torch.manual_seed(0)
    grads = [torch.randn(7) for _ in range(4)]

    def run(kinds):
        p = [torch.nn.Parameter(torch.linspace(-1, 1, 7))]
        opt, sd = None, None
        for k, g in zip(kinds, grads):
            new = (DenseAdam if k == "d" else torch.optim.Adam)(p, lr=1e-2)
            if sd is not None:
                new.load_state_dict(sd)
            opt = new
            p[0].grad = g.clone()
            opt.step()
            sd = opt.state_dict()
        return p[0].detach().clone()

    ref = run("tttt")
    for kinds in ("dddd", "dtdt", "tdtd", "ttdd"):
        torch.testing.assert_close(run(kinds), ref, rtol=0, atol=1e-7)


def test_dense_adam_state_dict_leaves_the_live_state_alone():
    """torch's Optimizer.state_dict() hands out the LIVE per-parameter dicts; DenseAdam leaves the
    derived 'active' flags out of the file without taking them away from the optimizer (a checkpoint in the middle of
    training used to delete them, and the captured trainer step reads them on the next call)."""
    from miso_amd.optim import DenseAdam
    p = torch.nn.Parameter(torch.zeros(5))
    opt = DenseAdam([p], lr=1e-2)
    p.grad = torch.ones(5)
    opt.step()
    flags = torch.ones(1, dtype=torch.uint8)
    opt.state[p]["active"] = flags                 # what the kernel path keeps next to the moments
    sd = opt.state_dict()
    assert "active" not in sd["state"][0] and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert opt.state[p]["active"] is flags
    assert sd["state"][0]["exp_avg"] is opt.state[p]["exp_avg"]        # still references, as torch's own state_dict


@pytest.mark.parametrize("tag", ["eik", "grad", "all", "allL2"])
def test_isdf_eikonal_gradient_smoothness_branches_match_reference(device_backend, tag, monkeypatch):
    """iSDFLoss.compute_default with the terms that differentiate a spatial gradient taken with create_graph=True
    (loss_isdf.py:99-150; configs/base.yaml:40 ships eik_weight 50): loss values, the gradient w.r.t. every feature
    level (through the SECOND backward of the encode: miso_encode_bwd2 / miso_grad_pull_dx on the GPU) and w.r.t.
    the coordinates, against tests/golden/second_order.npz -- the reference's own loss code run on the oracle's
    any-order sampling op (tools/make_goldens.py gen_second_order: its CUDA op cannot run in the build container)."""
    import miso_amd.grid_opt.loss_isdf as LI
    dev = device_backend
    case = gc.CASES["small"]
    g = G("second_order")
    net = make_gridnet(case, dev)
    net.unlock_feature()
    kw = {"eik": dict(eik_weight=50.0), "grad": dict(grad_weight=0.02),
          "all": dict(eik_weight=50.0, grad_weight=0.02, smooth_weight=0.1),
          "allL2": dict(eik_weight=50.0, grad_weight=0.02, smooth_weight=0.1, loss_type="L2")}[tag]
    il = LI.iSDFLoss("grid_net", trunc_weight=5.0, trunc_distance=0.15, eik_apply_dist=0.1, smooth_std=0.05,
                     slam_mode=False, **kw)
    mi = {"coords": T(g["coords"])[None].to(dev), "normals": T(g["normals"]).to(dev)}
    gt = {"sdf": T(g["bounds"]).to(dev), "grad_vec": T(g["grad_vec"]).to(dev)}
    noise = T(g["noise"]).to(dev)
    with monkeypatch.context() as m:
        m.setattr(torch, "randn_like", lambda t, *a, **k: noise.clone())     # the recorded draw of loss_isdf.py:143
        d = il.compute(net, mi, gt)
    sum(v.mean() for v in d.values()).backward()
    assert set(d) == ({"sdf", "smooth"} if "smooth_weight" in kw else {"sdf"})
    for k_, v in d.items():
        ref = float(g[f"isdf_{tag}_{k_}"])
        assert abs(v.item() - ref) <= 2e-5 * max(1.0, abs(ref)), (k_, v.item(), ref)
    for l in range(case["n_levels"]):
        close(net.features[l].feature.grad, T(g[f"isdf_{tag}_gfeat{l}"]), 2e-4, 1e-7)
    close(mi["coords"].grad, T(g[f"isdf_{tag}_gcoords"]), 2e-4, 1e-7)


@pytest.mark.parametrize("method", ["autograd", "finitediff"])
def test_mapping_eikonal_term_matches_reference(device_backend, method):
    """miso_loss_eikonal (loss.py:638-665) with both gradient methods -- 'autograd' goes through the second-order
    encode, 'finitediff' (what configs/rgbd/scannet.yaml:45-49 selects) through six extra forwards -- and
    MisoLossMapping.compute with weight_eik > 0 composed from it (the reference's own compute raises there:
    loss.py:788 reads an undefined self.use_clip)."""
    import miso_amd.grid_opt.loss as L
    dev = device_backend
    case = gc.CASES["small"]
    g = G("second_order")
    net = make_gridnet(case, dev)
    net.unlock_feature()
    x, sdf_t = T(g["coords"]).to(dev), T(g["eik_gt_sdf"]).to(dev)
    val = L.miso_loss_eikonal(model=net, coords_world=x, gt_sdf=sdf_t, eik_trunc_dist=0.1, grad_method=method,
                              finite_diff_eps=1e-2)
    val.backward()
    ref = float(g[f"eik_{method}"])
    assert abs(val.item() - ref) <= 2e-5 * abs(ref)
    grads = [net.features[l].feature.grad.clone() for l in range(case["n_levels"])]
    # central differences divide fp32 SDF values by 2 eps = 0.02: a 1e-8 difference between two fp32 evaluation orders
    # of the decoder becomes 5e-7 in the spatial gradient and enters every feature gradient
    rtol = 1e-3 if method == "finitediff" else 2e-4
    for l in range(case["n_levels"]):
        close(grads[l], T(g[f"eik_{method}_gfeat{l}"]), rtol, 1e-7)
    # through the loss class: sdf + weight_eik * eik (+ free space), one keyframe at the identity
    net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
    net.lock_pose()
    n = x.shape[0]
    mi = {"coords_frame": x[None], "sample_frame_ids": torch.zeros(1, n, 1, dtype=torch.int64, device=dev),
          "weights": torch.ones(1, n, 1, device=dev)}
    gt = {"sdf": sdf_t[None], "sdf_valid": torch.ones(1, n, 1, device=dev), "sdf_signs": torch.zeros(1, n, 1, device=dev)}
    lf = L.MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.5, weight_fs=0.0, trunc_dist=0.15,
                           eik_trunc_dist=0.1, grad_method=method, finite_diff_eps=1e-2)
    d = lf.compute(net, mi, gt)
    assert set(d) == {"sdf_L1", "eik"} and abs(d["eik"].item() - 0.5 * ref) <= 2e-5 * ref


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 70000])
def test_fast_captured_step_equals_the_checked_one(n, tmp_path):
    """GridTrainer's fast plan (batch written by one launch, optimizer inside the graph replay, step scalars and step
    count on the device, NaN guards resolved without waiting) against the checked captured path (fast_captured_step
    off): identical features / Adam moments / step counts over 12 steps of changing batches, one of which has a NaN
    label (the reference skips backward and optimizer.step(): nothing moves, the count does not advance); then the
    plan lets go when something it baked in changes (learning rate; a pose unlocked) and the results still agree;
    and the optimizer state round-trips through state_dict.  n = 4096: atomic scatter; 70000: binned path."""
    from miso_amd.grid_opt.loss import MisoLossMapping
    from miso_amd.grid_opt.models.grid_net import GridNet
    from miso_amd.grid_opt.trainer import GridTrainer
    dev = "cuda:0"
    c = gc.ATLAS
    # 16 x 8 x 16 and 64 x 32 x 64 vertices, 4 channels: both levels big enough for the optimizer's kernel path
    cfg = gc.model_cfg(c["bound"], 0.25, 4, 2, c["fdim"], c["hidden"], num_poses=2, init_stddev=1e-2)
    g = torch.Generator().manual_seed(2)
    batches = []
    for b in range(4):
        x = (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([1.8, 0.9, 1.8]) * (0.5 + 0.15 * b)
        mi = {"coords_frame": x[None].to(dev), "sample_frame_ids": torch.randint(0, 2, (1, n, 1), generator=g).to(dev),
              "weights": (torch.rand(1, n, 1, generator=g) + 0.5).to(dev)}
        gt = {"sdf": (torch.rand(1, n, 1, generator=g) * 0.2 - 0.1).to(dev),
              "sdf_valid": (torch.rand(1, n, 1, generator=g) > 0.1).float().to(dev),
              "sdf_signs": (torch.rand(1, n, 1, generator=g) > 0.6).float().to(dev)}
        batches.append((mi, gt))
    bad = ({k: v.clone() for k, v in batches[1][0].items()}, {k: v.clone() for k, v in batches[1][1].items()})
    bad[1]["sdf"][0, 7, 0] = float("nan")

    def run(fast):
        torch.manual_seed(0)
        net = GridNet(cfg, device=dev).to(dev)
        R1 = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
        net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
        net.set_initial_kf_pose(1, R1, torch.tensor([[0.1], [-0.05], [0.02]]), kf_key="KF1")
        net.unlock_feature()
        net.lock_pose()
        tcfg = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-2, "epochs": 1, "ckpt_every": -1,
                "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": str(tmp_path),
                "relchange_tol": 0, "max_epochs_in_level": 1000, "grid_training_mode": "joint",
                "fast_captured_step": fast}
        lf = MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
        tr = GridTrainer(tcfg, net, lf, None, None, dev, torch.float32)
        losses, used_fast = [], 0
        for it in range(12):
            mi, gt = bad if it == 6 else batches[it % 4]
            losses.append(tr.train_step(mi, gt))
            used_fast += tr.__dict__.get("_fast_plan") is not None
            if it == 8:
                # a checkpoint in the middle of training (Trainer.save_model from post_epoch): the file carries no
                # 'active' flags, the optimizer keeps them, and the fast plan keeps replaying
                mid = tr.optimizer.state_dict()
                assert all("active" not in st for st in mid["state"].values())
                assert all("active" in tr.optimizer.state[f.feature] for f in net.features)
        torch.cuda.synchronize()
        feats = [f.feature for f in net.features]
        snap = lambda: ([f.detach().clone() for f in feats]
                        + [tr.optimizer.state[f][k].clone() for f in feats for k in ("exp_avg", "exp_avg_sq")])
        out = {"after12": snap(), "losses": torch.stack([l.detach().reshape(()) for l in losses]).cpu()}
        tr.optimizer.resolve_guard()
        out["steps12"] = [tr.optimizer.state[f]["step"] for f in feats]
        out["skipped"] = tr.optimizer.skipped_steps
        out["used_fast"] = used_fast
        # a change the plan baked in: the learning rate
        for grp in tr.optimizer.param_groups:
            grp["lr"] = 3e-3
        for it in range(3):
            tr.train_step(*batches[it])
        out["after_lr"] = snap()
        # state dict round trip, then on
        sd = tr.optimizer.state_dict()
        tr.optimizer.load_state_dict(sd)
        for it in range(3):
            tr.train_step(*batches[it])
        tr.optimizer.resolve_guard()
        out["after_reload"] = snap()
        out["steps_end"] = [tr.optimizer.state[f]["step"] for f in feats]
        return out

    a, b = run(False), run(True)
    assert a["used_fast"] == 0 and b["used_fast"] >= 9
    assert a["steps12"] == b["steps12"] == [11] * len(a["steps12"]) and a["skipped"] == b["skipped"] == 1
    assert a["steps_end"] == b["steps_end"] == [17] * len(a["steps_end"])
    assert torch.isnan(a["losses"][6]) and torch.isnan(b["losses"][6])
    fin = torch.ones(12, dtype=torch.bool)
    fin[6] = False
    torch.testing.assert_close(a["losses"][fin], b["losses"][fin], rtol=2e-5, atol=1e-8)
    # The binned batch's order inside a tile differs from run to run (300 of 300 reruns), so gradients differ by ~3e-7
    # of their maximum; Adam's normalisation turns that into whole steps for the few elements whose gradient is of the
    # size of eps (measured, any path against itself: about one run in ten has ~1e-4 of the elements off by up to
    # 1.3 lr).  A wrong step count, table row, flag or stale buffer would move every trained element instead.
    for key in ("after12", "after_lr", "after_reload"):
        for ta, tb in zip(a[key], b[key]):
            scale = ta.abs().max().item()
            d = (ta - tb).abs()
            assert d.mean().item() <= 2e-5 * scale and (d > 2e-4 * scale).float().mean().item() <= 2e-3, \
                (key, d.max().item(), d.mean().item(), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("lt", ["GM", "L2"])
def test_lm_step_on_device_equals_the_op_by_op_step(lt, monkeypatch):
    """Tracker.lm_step as one library call (miso_lm_track_step: one host synchronisation) against the op-by-op version
    (eight synchronisations): four consecutive steps from the same start give the same info and the same pose
    corrections, with the truncation filter active, a bool validity mask, label columns that are strided views, and
    samples outside the bound; wrong frame ids / invalid rows raise the reference's assertions and leave the pose
    alone."""
    from miso_amd.grid_opt.slam.tracker import Tracker
    dev = "cuda:0"
    case = gc.CASES["small"]
    g = G("tracker")
    pts = T(gc.make_points(case))
    n = pts.shape[0]
    gen = torch.Generator().manual_seed(3)
    pts = pts * 1.15                                            # some leave the bound
    pts[3, 1] = float("nan")                                    # get_batch's nan_to_num: folded into the fused step
    pts[9, 0] = float("inf")
    lab = torch.stack([T(g["sdf"])[:, 0] * (1.0 + 0.5 * torch.rand(n, generator=gen)), torch.ones(n), torch.zeros(n),
                       torch.ones(n)], dim=1).to(dev)            # (n,4) block: columns are strided views
    lab[5, 0] = float("nan")

    def make(valid_dtype=torch.bool, frame=1, bad_valid=False):
        net = make_gridnet(case, dev, num_poses=2, optimize_pose=True)
        net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
        net.set_initial_kf_pose(1, T(g["R0"]), T(g["t0"]), kf_key="KF1")
        valid = (lab[:, 1:2] > 0) if valid_dtype == torch.bool else lab[:, 1:2]
        if bad_valid:
            valid = valid.clone()
            valid[5] = 0
        fid = torch.full((n, 1), frame, dtype=torch.int64, device=dev)

        class DS(torch.utils.data.Dataset):
            def select_keyframes(self, kfs):
                pass

            def __len__(self):
                return 1

            def __getitem__(self, i):
                return ({"coords_frame": pts.to(dev), "sample_frame_ids": fid, "weights": lab[:, 3:4]},
                        {"sdf": lab[:, 0:1], "sdf_valid": valid, "sdf_signs": lab[:, 2:3]})

        cfg = {"device": dev, "train": {},
               "tracking": {"learning_rate": 1e-3, "verbose": False, "gm_scale_sdf": 0.1, "lm_lambda": 5.0,
                            "lm_max_iter": 3, "lm_tol_deg": 0.0, "lm_tol_m": 0.0, "loss_type": lt,
                            "trunc_dist": 0.12, "solver": "lm"}}
        return net, Tracker(net, DS(), cfg)

    res = {}
    for fused in (False, True):
        net, trk = make()
        if not fused:
            monkeypatch.setattr(Tracker, "_lm_step_on_device", lambda self, *a: None)
        else:
            monkeypatch.undo()
        infos = [trk.lm_step(1) for _ in range(4)]
        assert (trk.__dict__.get("_lm_dev") is not None) == fused
        res[fused] = (infos, net.rotation_corrections.detach().clone(), net.translation_corrections.detach().clone())
    for a, b in zip(res[False][0], res[True][0]):
        for key in a:
            assert abs(a[key] - b[key]) <= 2e-4 * abs(a[key]) + 1e-7, (key, a[key], b[key])
    assert 0.0 < res[True][0][0]["fov_overlap"] < 1.0
    close(res[True][1], res[False][1], 2e-4, 1e-7)
    close(res[True][2], res[False][2], 2e-4, 1e-7)
    # the reference's assertions, without having moved the pose
    for kw in (dict(frame=0), dict(bad_valid=True)):
        net, trk = make(**kw)
        before = (net.rotation_corrections.detach().clone(), net.translation_corrections.detach().clone())
        with pytest.raises(AssertionError):
            trk.lm_step(1)
        assert torch.equal(net.rotation_corrections.detach(), before[0])
        assert torch.equal(net.translation_corrections.detach(), before[1])


@pytest.mark.gpu
@pytest.mark.parametrize("lt", ["L1", "L2", "GM"])
def test_track_window_on_device_equals_the_trainer_window(lt, monkeypatch, tmp_path, caplog):
    """Tracker.track_window (the 'adam' solver of configs/rgbd/scannet.yaml) as library calls without a host round trip
    (miso_track_adam_step) against the Trainer + MisoLossTracking + autograd + DenseAdam window it replaces: the same
    pose corrections after 8 iterations from the same start, with a truncation filter, a bool validity mask with holes,
    strided label columns and samples outside the bound; other keyframes' corrections do not move; a NaN loss (NaN
    features) skips every step and says so."""
    import logging
    from miso_amd.grid_opt.slam.tracker import Tracker
    dev = "cuda:0"
    case = gc.CASES["small"]
    g = G("tracker")
    pts = (T(gc.make_points(case)) * 1.1).to(dev)
    n = pts.shape[0]
    gen = torch.Generator().manual_seed(4)
    lab = torch.stack([T(g["sdf"])[:, 0] * (1.0 + 0.5 * torch.rand(n, generator=gen)),
                       (torch.rand(n, generator=gen) > 0.1).float()], dim=1).to(dev)
    fid = torch.ones(n, 1, dtype=torch.int64, device=dev)

    def run(fused, nan_feats=False):
        net = make_gridnet(case, dev, num_poses=3, optimize_pose=True)
        net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
        net.set_initial_kf_pose(1, T(g["R0"]), T(g["t0"]), kf_key="KF1")
        net.set_initial_kf_pose(2, torch.eye(3), torch.ones(3, 1) * 0.1, kf_key="KF2")
        with torch.no_grad():
            net.rotation_corrections[1] += torch.tensor([0.02, -0.01, 0.015], device=dev)
            net.rotation_corrections[2] += 0.3
        if nan_feats:
            with torch.no_grad():
                net.features[0].feature.fill_(float("nan"))

        class DS(torch.utils.data.Dataset):
            def select_keyframes(self, kfs):
                pass

            def __len__(self):
                return 1

            def __getitem__(self, i):
                sdf = lab[:, 0:1]
                return ({"coords_frame": pts, "sample_frame_ids": fid, "weights": torch.ones(n, 1, device=dev)},
                        {"sdf": sdf, "sdf_valid": lab[:, 1:2] > 0, "sdf_signs": torch.zeros(n, 1, device=dev)})

        cfg = {"device": dev,
               "train": {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1, "ckpt_every": -1,
                         "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": str(tmp_path)},
               "tracking": {"learning_rate": 2e-3, "verbose": False, "gm_scale_sdf": 0.1, "lm_lambda": 5.0,
                            "lm_max_iter": 3, "lm_tol_deg": 0.0, "lm_tol_m": 0.0, "loss_type": lt,
                            "trunc_dist": 0.15, "solver": "adam"}}
        trk = Tracker(net, DS(), cfg)
        if not fused:
            monkeypatch.setattr(Tracker, "_track_window_on_device", lambda self, *a: False)
        else:
            monkeypatch.undo()
        trk.track_window([1], iterations=8)
        assert (trk.__dict__.get("_adam_dev") is not None) == fused
        return net.rotation_corrections.detach().clone(), net.translation_corrections.detach().clone()

    ref, got = run(False), run(True)
    assert (got[0][1] - torch.tensor([0.02, -0.01, 0.015], device=dev)).abs().max() > 5e-3      # it did move
    # Adam normalises the gradient: after 8 steps of 2e-3 the two paths may differ by a fraction of one step
    assert (ref[0] - got[0]).abs().max().item() <= 2e-4 and (ref[1] - got[1]).abs().max().item() <= 2e-4
    for k in (0, 2):                                   # the locked keyframes
        assert torch.equal(got[0][k], ref[0][k]) and torch.equal(got[1][k], ref[1][k])
    with caplog.at_level(logging.WARNING):
        ref_n, got_n = run(False, nan_feats=True), run(True, nan_feats=True)
    assert sum("Loss is nan" in r.getMessage() for r in caplog.records) == 16          # 8 skipped steps each
    assert torch.equal(ref_n[0], got_n[0]) and torch.equal(ref_n[1], got_n[1])          # nothing moved
    assert torch.allclose(got_n[0][1], torch.tensor([0.02, -0.01, 0.015], device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["coordinate+joint", "joint"])
def test_mapper_calls_adopt_the_fast_plan(mode, tmp_path):
    """The SLAM loop builds a new GridTrainer (new optimizers, fresh Adam state) for every Mapper.mapping call.  With
    fast_captured_step the plan an earlier call left with the model is adopted by the next trainer -- its optimizer's
    state becomes the plan's zeroed buffers -- instead of paying for a new step, capture and plan per call.  Three calls
    of 8 iterations (coordinate schedule: 3 per level, then joint) must leave the same features as the checked path,
    which builds everything anew each time; and from the second call on every step but none must go through a plan."""
    from miso_amd.grid_opt.slam.mapper import Mapper
    import miso_amd.grid_opt.trainer as TR
    dev = "cuda:0"
    c = gc.ATLAS
    cfg_m = gc.model_cfg(c["bound"], 0.25, 4, 2, c["fdim"], c["hidden"], num_poses=2, init_stddev=1e-2)
    n = 20000
    g = torch.Generator().manual_seed(6)
    batches = []
    for b in range(3):
        x = ((torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([1.8, 0.9, 1.8])).to(dev)
        batches.append(({"coords_frame": x, "sample_frame_ids": torch.randint(0, 2, (n, 1), generator=g).to(dev),
                         "weights": torch.ones(n, 1, device=dev)},
                        {"sdf": (torch.rand(n, 1, generator=g) * 0.2 - 0.1).to(dev), "sdf_valid": torch.ones(n, 1, device=dev),
                         "sdf_signs": (torch.rand(n, 1, generator=g) > 0.5).float().to(dev)}))

    def run(fast):
        torch.manual_seed(0)
        from miso_amd.grid_opt.models.grid_net import GridNet
        net = GridNet(cfg_m, device=dev).to(dev)
        net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
        net.set_initial_kf_pose(1, torch.eye(3), torch.tensor([[0.1], [0.0], [-0.05]]), kf_key="KF1")
        state = {"i": 0}

        class DS(torch.utils.data.Dataset):
            def select_keyframes(self, kfs):
                pass

            def __len__(self):
                return 1

            def __getitem__(self, i):
                state["i"] += 1
                return batches[state["i"] % 3]

        cfg = {"device": dev,
               "train": {"trainer": "base", "verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 50,
                         "ckpt_every": -1, "eval_every": -1, "pretrained_model": None, "log_dir": str(tmp_path),
                         "relchange_tol": 0, "max_epochs_in_level": 100, "grid_training_mode": mode,
                         "fast_captured_step": fast},
               "mapping": {"learning_rate": 5e-3, "loss_type": "L1", "weight_sdf": 1.0, "weight_eik": 0.0, "weight_fs": 0.1,
                           "trunc_dist": 0.15, "finite_diff_eps": 0.01, "grad_method": "finitediff",
                           "eik_trunc_dist": 0.024, "verbose": False}}
        mp = Mapper(net, DS(), cfg)
        runs = []
        orig = TR._FastMappingPlan.run

        def counting(self, *a, **k):
            out = orig(self, *a, **k)
            runs.append(out is not None)
            return out

        TR._FastMappingPlan.run = counting
        try:
            per_call = []
            for _ in range(3):
                before = sum(runs)
                mp.mapping([0, 1], iterations=8, level_iterations=3)
                per_call.append(sum(runs) - before)
        finally:
            TR._FastMappingPlan.run = orig
        torch.cuda.synchronize()
        return [f.feature.detach().clone() for f in net.features], per_call

    ref, calls_ref = run(False)
    got, calls = run(True)
    assert calls_ref == [0, 0, 0]
    assert calls[1] == 8 and calls[2] == 8, calls            # every step of the later calls through an adopted plan
    # 24 Adam steps of 5e-3 each.  Typically the two runs agree to 5e-8; now and then a handful of elements whose
    # gradient is within rounding of zero part by whole steps (measured: 6 of 524 288 elements by 2 x lr) -- the binned
    # batch's order inside a tile is not reproducible, Adam's normalisation turns the sign of a 1e-12 gradient into a
    # full step.  A wrong state adoption (stale moments, flags, step count) moves every trained element instead.
    for a, b in zip(ref, got):
        d = (a - b).abs()
        assert d.mean().item() <= 5e-6 and (d > 1e-5).float().mean().item() <= 2e-3, (d.max().item(), d.mean().item())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["joint", "coordinate+joint"])
def test_fast_plan_on_the_headline_grid(mode, tmp_path):
    """The one-replay plan on the cfg-2 grid (3 levels {32,64,128}^3, 8 channels, all levels pulled by the block kernel)
    with 262 144 samples, joint and on the coordinate schedule (2 epochs per level): same features as the checked path
    after 8 steps (noise-aware bound, see test_fast_captured_step_equals_the_checked_one)."""
    from miso_amd.grid_opt.loss import MisoLossMapping
    from miso_amd.grid_opt.models.grid_net import GridNet
    from miso_amd.grid_opt.trainer import GridTrainer
    dev = "cuda:0"
    cfg = gc.model_cfg([[-1.0, 1.0]] * 3, 2.0 / 32, 2, 3, 8, 64, num_poses=1, init_stddev=1e-2)
    n = 262144
    g = torch.Generator().manual_seed(12)
    x = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)
    mi = {"coords_frame": x[None], "sample_frame_ids": torch.zeros(1, n, 1, dtype=torch.int64, device=dev),
          "weights": torch.ones(1, n, 1, device=dev)}
    gt = {"sdf": (torch.rand(1, n, 1, generator=g) * 0.2 - 0.1).to(dev), "sdf_valid": torch.ones(1, n, 1, device=dev),
          "sdf_signs": (torch.rand(1, n, 1, generator=g) > 0.7).float().to(dev)}

    def run(fast):
        torch.manual_seed(0)
        net = GridNet(cfg, device=dev).to(dev)
        net.set_initial_kf_pose(0, torch.eye(3), torch.zeros(3, 1), kf_key="KF0")
        net.unlock_feature()
        net.lock_pose()
        tcfg = {"verbose": False, "optimizer": "adam", "learning_rate": 1e-3, "epochs": 1, "ckpt_every": -1,
                "eval_every": -1, "eval_metric": None, "pretrained_model": None, "log_dir": str(tmp_path),
                "relchange_tol": 0, "max_epochs_in_level": 2, "grid_training_mode": mode, "fast_captured_step": fast}
        lf = MisoLossMapping(loss_type="L1", weight_sdf=1.0, weight_eik=0.0, weight_fs=0.1, trunc_dist=0.15)
        tr = GridTrainer(tcfg, net, lf, None, None, dev, torch.float32)
        used = 0
        for epoch in range(8):
            tr.pre_epoch(epoch)
            tr.train_step(mi, gt)
            used += tr.__dict__.get("_fast_plan") is not None
        torch.cuda.synchronize()
        return [f.feature.detach().clone() for f in net.features], used

    ref, u0 = run(False)
    got, u1 = run(True)
    assert u0 == 0 and u1 >= (5 if mode == "joint" else 1)
    for a, b in zip(ref, got):
        d = (a - b).abs()
        assert d.mean().item() <= 1e-6 and (d > 1e-5).float().mean().item() <= 2e-3, (d.max().item(), d.mean().item())
    assert not torch.equal(ref[2], torch.zeros_like(ref[2]))
