"""Kernel-level timing of the hot path at BASELINE cfg-2 (dev tool, not the contract bench)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    dev = "cuda:0"
    n = int(os.environ.get("N", 262144))
    L, C, H = 3, 8, 64
    sizes = [32, 64, 128]
    torch.manual_seed(0)
    feats = [(torch.randn(1, C, s, s, s, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
             for s in sizes]
    meta = ops.GridMeta.from_bound([[-1.0, 1.0]] * 3)
    lin = [torch.nn.Linear(L * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pack = ops.DecoderPack([l.weight.data.to(dev) for l in lin], [l.bias.data.to(dev) for l in lin])
    g = torch.Generator().manual_seed(1234)
    x = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)
    xs = x[torch.argsort((x[:, 2] * 64).floor() * 1e6 + (x[:, 1] * 64).floor() * 1e3 + (x[:, 0] * 64).floor())].contiguous()
    gs = torch.randn(n, 1, device=dev) / n
    grads = [torch.zeros_like(f) for f in feats]
    B = 20 + 64 * L * C
    for tag, xx in (("random", x), ("sorted", xs)):
        sdf, mask = ops.sdf_fwd_raw(xx, feats, meta, pack, True)
        t_f = timeit(lambda: ops.sdf_fwd_raw(xx, feats, meta, pack, True))
        t_fi = timeit(lambda: ops.sdf_fwd_raw(xx, feats, meta, pack, False))
        t_b = timeit(lambda: ops.sdf_bwd_raw(xx, feats, meta, pack, gs, mask, False, [True] * L, grads))
        t_bx = timeit(lambda: ops.sdf_bwd_raw(xx, feats, meta, pack, gs, mask, True, [False] * L, None))
        t_z = timeit(lambda: [g_.zero_() for g_ in grads])
        t_e = timeit(lambda: ops.encode_fwd_raw(xx, feats, meta))
        go = torch.randn(n, L * C, device=dev)
        t_eb = timeit(lambda: ops.encode_bwd_raw(xx, feats, meta, go, False, [True] * L))
        sb = ops.SortedBatch(n, dev)
        t_s = timeit(lambda: sb.sort(xx, meta))
        _, mask_s = ops.sdf_fwd_raw(xx, feats, meta, pack, True, sorted_batch=sb)
        t_fs = timeit(lambda: ops.sdf_fwd_raw(xx, feats, meta, pack, True, mask=mask_s, sorted_batch=sb))
        t_bs = timeit(lambda: ops.sdf_bwd_raw(xx, feats, meta, pack, gs, mask_s, False, [True] * L, grads, sorted_batch=sb, overwrite=True))
        print(f"[{tag}] binned: sort {t_s:.1f}us fwd {t_fs:.1f}us bwd(grid) {t_bs:.1f}us -> step "
              f"{t_s + t_fs + t_bs:.1f}us = {n / (t_s + t_fs + t_bs):.1f} Mpts/s, frac "
              f"{n * B / ((t_s + t_fs + t_bs) * 1e-6) / 8e12:.3f}")
        tot = t_f + t_b + t_z
        print(f"[{tag}] N={n} fwd(train) {t_f:.1f}us fwd(infer) {t_fi:.1f}us bwd(grid) {t_b:.1f}us "
              f"bwd(x only) {t_bx:.1f}us zero {t_z:.1f}us | encode {t_e:.1f}us enc_bwd(+alloc) {t_eb:.1f}us")
        print(f"   fwd+bwd+zero = {tot:.1f}us -> {n / tot:.1f} Mpts/s, HBM-roofline frac "
              f"{n * B / (tot * 1e-6) / 8e12:.3f}")
    p = feats[2]
    m, v, gd = torch.zeros_like(p), torch.zeros_like(p), torch.randn_like(p)
    t_a = timeit(lambda: ops.adam_dense_(p, gd, m, v, 3, 1e-3))
    print(f"adam 128^3x8 ({p.numel() * 28 / 1e6:.0f} MB traffic): {t_a:.1f}us -> {p.numel() * 28 / t_a / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
