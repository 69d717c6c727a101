"""Dev tool: dense SDF extraction (utils_sdf.extract_fields) at resolution 256 / 512 on the cfg-2 grid."""
import sys, time, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import golden_cases as gc
from miso_amd.grid_opt.models.grid_net import GridNet
from miso_amd.grid_opt.utils.utils_sdf import extract_fields
dev = 'cuda:0'
cfg = gc.model_cfg([[-1.0, 1.0]] * 3, 2.0 / 32, 2, 3, 8, 64, init_stddev=1e-2)
torch.manual_seed(0)
net = GridNet(cfg, device=dev).to(dev)
lo, hi = torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0])
for res in (128, 256, 512):
    f = lambda pts: net(pts)
    extract_fields(lo, hi, res, f, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    u = extract_fields(lo, hi, res, f, device=dev)
    dt = time.perf_counter() - t0
    print(f"res {res}: {dt * 1e3:.1f} ms ({res ** 3 / dt / 1e9:.2f} G pts/s incl. the copy of the volume to the host), shape {u.shape}")
