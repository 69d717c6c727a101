"""Dev bench: marching cubes on a res^3 volume in HBM (sphere + ripple).  Prints per-launch times (HIP events on
the launch stream) and the HBM rate of the two sweeps against their algorithmic bytes (4 B per sample each)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda:0"
    lib = _lib.load()
    out = {}
    for n in (256, 512):
        ax = torch.arange(n, device=dev, dtype=torch.float32)
        x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
        c = n / 2 - 0.7
        u = torch.sqrt((x - c) ** 2 + (y - c + 1.2) ** 2 + (z - c - 0.9) ** 2) - 0.35 * n \
            + 0.8 * torch.sin(0.21 * x) * torch.cos(0.17 * y)
        del x, y, z
        u = u.contiguous()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        W = lib.miso_mc_words(n, n, n)
        ws_bytes = lib.miso_mc_workspace_bytes(n, n, n)
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=dev)
        counts = torch.empty(4 * W + 2, dtype=torch.int32, device=dev)
        offs = torch.empty(4 * W, dtype=torch.int64, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())
        t_cls = timeit(lambda: lib.miso_mc_classify(P(u), n, n, n, 0.0, P(ws), P(counts), st))

        def scan():
            torch.cumsum(counts[:3 * W], 0, dtype=torch.int64, out=offs[:3 * W])
            torch.cumsum(counts[3 * W:4 * W], 0, dtype=torch.int64, out=offs[3 * W:])
            offs.sub_(counts[:4 * W])

        t_scan = timeit(scan)
        scan()
        nv = int(offs[3 * W - 1] + counts[3 * W - 1])
        nt = int(offs[4 * W - 1] + counts[4 * W - 1])
        tch, vch = int(counts[4 * W]), int(counts[4 * W + 1])
        faces = torch.empty((nt, 3), dtype=torch.int64, device=dev)
        verts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        t_emit = timeit(lambda: lib.miso_mc_emit(n, n, n, P(ws), P(offs), tch, nt, P(faces), st))
        t_verts = timeit(lambda: lib.miso_mc_vertices(P(u), n, n, n, 0.0, P(ws), P(offs), vch, nv, P(verts), st))
        t_all = timeit(lambda: ops.marching_cubes(u, 0.0), iters=10, warm=2)
        vol_bytes = 4 * n ** 3
        cls_bytes = vol_bytes + ws_bytes            # volume read once, workspace written once
        out[f"res{n}"] = {"triangles": nt, "vertices": nv, "chunks": W, "chunks_with_triangles": tch, "chunks_with_vertices": vch, "workspace_MB": ws_bytes / 1e6,
                          "classify_us": t_cls, "classify_GBps": cls_bytes / t_cls / 1e3,
                          "scan_us": t_scan, "emit_us": t_emit,
                          "vertices_us": t_verts, "marching_cubes_total_us": t_all}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
