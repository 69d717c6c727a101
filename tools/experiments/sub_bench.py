#!/usr/bin/env python3
"""grad_pull_sub_kernel alone at the ScanNet fine level (dev tool): 200 x 100 x 200, C=4, 540 000 samples crowding the
middle of the bound; phases by ablation (MISO_DEBUG_PULL: 4 no sweep, 16 no pull)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402
from tools.quick_bench import timeit  # noqa: E402


def main():
    os.environ.setdefault("MISO_PULL_SUB", "1")      # the kernel is opt-in
    dev = "cuda:0"
    n = int(os.environ.get("N", 540000))
    C = 4
    torch.manual_seed(0)
    feats = [(torch.randn(1, C, 200, 100, 200, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)]
    meta = ops.GridMeta.from_bound([[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]])
    g = torch.Generator().manual_seed(1)
    x = ((torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([6.0, 2.5, 6.0])).to(dev)
    if os.environ.get("UNIFORM"):
        x = ((torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([10.0, 5.0, 10.0])).to(dev)
    sb = ops.SortedBatch(n, dev).sort(x, meta)
    df = torch.randn(n, C, device=dev)
    grads = [torch.empty_like(feats[0])]
    out = []
    for dbg in ("0", "4", "16"):
        os.environ["MISO_DEBUG_PULL"] = dbg
        t = timeit(lambda: ops.grad_pull_raw(feats, meta, sb, df, grads, overwrite=True))
        out.append(f"debug {dbg}: {t:.1f} us")
    os.environ["MISO_DEBUG_PULL"] = "0"
    print(" | ".join(out))


if __name__ == "__main__":
    main()
