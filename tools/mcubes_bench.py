#!/usr/bin/env python3
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
        nb = lib.miso_mc_blocks(n, n, n)
        counts = torch.empty(nb, dtype=torch.int32, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())
        t_count = timeit(lambda: lib.miso_mc_count(P(u), n, n, n, 0.0, P(counts), st))
        incl = torch.cumsum(counts, 0, dtype=torch.int64)
        offs = (incl - counts).contiguous()
        nt = int(incl[-1])
        keys = torch.empty((nt, 3), dtype=torch.int64, device=dev)
        t_emit = timeit(lambda: lib.miso_mc_emit(P(u), n, n, n, 0.0, P(offs), nt, P(keys), st))
        t_unique = timeit(lambda: torch.unique(keys.view(-1), sorted=True, return_inverse=True), iters=5, warm=1)
        uniq, inv = torch.unique(keys.view(-1), sorted=True, return_inverse=True)
        verts = torch.empty((uniq.shape[0], 3), dtype=torch.float32, device=dev)
        t_verts = timeit(lambda: lib.miso_mc_vertices(P(u), n, n, n, 0.0, P(uniq), uniq.shape[0], P(verts), st))
        t_all = timeit(lambda: ops.marching_cubes(u, 0.0), iters=5, warm=1)
        vol_bytes = 4 * n ** 3
        out[f"res{n}"] = {"triangles": nt, "vertices": int(uniq.shape[0]),
                          "count_us": t_count, "count_GBps": vol_bytes / t_count / 1e3,
                          "emit_us": t_emit, "emit_GBps": (vol_bytes + 24 * nt) / t_emit / 1e3,
                          "unique_us": t_unique, "vertices_us": t_verts, "marching_cubes_total_us": t_all}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
