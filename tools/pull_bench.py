"""grad_pull_kernel alone at cfg-2 (dev tool): all levels together, each level alone, pairs of levels, and the sort."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402
from tools.quick_bench import timeit  # noqa: E402


def main():
    dev = "cuda:0"
    n = int(os.environ.get("N", 262144))
    L, C = 3, 8
    sizes = [32, 64, 128]
    torch.manual_seed(0)
    feats = [(torch.randn(1, C, s, s, s, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
             for s in sizes]
    meta = ops.GridMeta.from_bound([[-1.0, 1.0]] * 3)
    g = torch.Generator().manual_seed(1234)
    x = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)
    sb = ops.SortedBatch(n, dev).sort(x, meta)
    df = torch.randn(n, L * C, device=dev)
    grads = [torch.empty_like(f) for f in feats]
    t_s = timeit(lambda: sb.sort(x, meta))
    out = [f"sort {t_s:.1f}"]
    masks = [int(m, 2) for m in os.environ.get("MASKS", "111,001,010,100,011,110").split(",")]
    for mask in masks:
        gl = [grads[l] if (mask >> l) & 1 else None for l in range(L)]
        t = timeit(lambda: ops.grad_pull_raw(feats, meta, sb, df, gl, overwrite=True))
        out.append(f"levels {mask:03b}: {t:.1f}")
    print(" | ".join(out))


if __name__ == "__main__":
    main()
