"""Dev tool: where the time of PosedSdfRgbd.__getitem__ goes (ScanNet knobs)."""
import sys, time, torch
sys.path.insert(0, '.')
from miso_amd import ops
from miso_amd.grid_opt.datasets.sdf_rgbd import PosedSdfRgbd
from miso_amd.grid_opt.utils.utils_data import CameraParameters
from miso_amd.grid_opt.utils.utils_sample import sample_pixels
dev = 'cuda:0'
B, H, W, rays, n_strat, n_surf = 100, 480, 640, 200, 19, 8
g = torch.Generator().manual_seed(3)
depth = torch.rand(B, H, W, generator=g) * 4.0 + 0.5
depth[torch.rand(B, H, W, generator=g) < 0.1] = 0.0
Rm = torch.eye(3).repeat(B, 1, 1)
t = torch.rand(B, 3, 1, generator=g) * 10 - 5
cp = CameraParameters(fx=577.6, fy=578.7, cx=318.9, cy=242.7, H=H, W=W)
ds = PosedSdfRgbd.from_frames(depth, Rm, t, cp, n_rays=rays, n_strat_samples=n_strat, n_surf_samples=n_surf,
                              trunc_dist=0.15, device=dev, normals=torch.ones(B, H, W, 3))

def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e6:9.1f} us")

n = B * rays
timeit("sample_pixels", lambda: sample_pixels(rays, B, H, W, dev))
timeit("rand u", lambda: torch.rand(n, n_strat, device=dev))
timeit("randn g", lambda: torch.randn(n, n_surf - 1, device=dev) * 0.1)
timeit("RayBatch alloc", lambda: ops.RayBatch(n, 27, dev))
out = ops.RayBatch(n, 27, dev)
timeit("sample_batch(out=)", lambda: ds.sample_batch(out=out))
timeit("sample_batch()", lambda: ds.sample_batch())
timeit("rows()", lambda: out.rows())
timeit("ds[0]", lambda: ds[0])
ds.select_keyframes(list(range(0, 100, 2)))
timeit("ds[0] 50 selected", lambda: ds[0])
