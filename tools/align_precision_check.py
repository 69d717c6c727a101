"""dev: pose gradients of one level-1 pair (4 M vertices): HIP fused plan vs the oracle loop in fp32 and in fp64."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import scannet_atlas
import oracle_backend
from miso_amd import ops
import miso_amd.grid_opt.align.miso as AM
dev = "cuda:0"
atlas = scannet_atlas(dev, 4)
atlas.precompute_coordinates_for_alignment()
S = atlas.num_submaps
R0 = torch.stack(list(atlas.R_world_submap_list)); t0 = torch.stack(list(atlas.t_world_submap_list))
prm0 = torch.cat((torch.cat([p.detach().reshape(1, 3) for p in atlas.rotation_corrections]),
                  torch.cat([p.detach().reshape(1, 3) for p in atlas.translation_corrections])), 1)
pairs = [(0, 1), (2, 3)]
level = int(os.environ.get("LEVEL", "1"))
inp = AM.latent_pair_inputs(atlas, pairs, level=level, fdim=4, check_intersection=True)
kw = dict(loss_type="L2", align_weight=3000.0, lr=1e-2, ring_iters=1)
plan = ops.AlignPlan(R0, t0, inp, **kw); plan.params.copy_(prm0); plan.iteration_a()
f_gpu = plan.flat.cpu().double()
def host(dt):
    out = []
    for pr in inp:
        q = dict(pr)
        for k in ("coords", "feats_src", "gate_pts"):
            q[k] = pr[k].detach().cpu().to(dt)
        q["feats_dst"] = [f.detach().cpu().contiguous().to(dt) for f in pr["feats_dst"]]
        out.append(q)
    ref = oracle_backend.AlignPlan(R0.cpu().to(dt), t0.cpu().to(dt), out, **kw)
    ref.params = prm0.cpu().to(dt); ref.flat = ref.flat.to(dt); ref.pair_losses = ref.pair_losses.to(dt)
    ref.iteration_a()
    return ref.flat.double()
f32 = host(torch.float32); f64 = host(torch.float64)
sc = f64[:-1].abs().max()
print("scale", sc.item(), "loss", f_gpu[-1].item(), f32[-1].item(), f64[-1].item())
print("hip  vs fp64:", ((f_gpu - f64)[:-1].abs().max() / sc).item())
print("cpu32 vs fp64:", ((f32 - f64)[:-1].abs().max() / sc).item())
print("hip  vs cpu32:", ((f_gpu - f32)[:-1].abs().max() / sc).item())
