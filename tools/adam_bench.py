"""Dense Adam over one level (dev): miso_adam_active (finds the moving chunks by reading the gradient) against
miso_adam_touched (reads the flags the scatter kernels left), by level size and fraction of moving chunks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402

dev = "cuda:0"
for numel, name in ((1_152_000, "cfg-5 coarse 120x120x20x4"), (16_777_216, "cfg-2 fine 128^3x8"),
                    (16_000_000, "cfg-3 fine 200x100x200x4"), (144_000_000, "cfg-5 fine 600x600x100x4")):
    for frac in (0.02, 0.3, 1.0):
        p = torch.randn(numel, device=dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        ch = ops._lib.ADAM_CHUNK                      # floats per flag
        nch = (numel + ch - 1) // ch
        on = (torch.rand(nch, device=dev) < frac)
        g = torch.randn(numel, device=dev) * on.repeat_interleave(ch)[:numel]
        res = []
        for kind in ("active", "touched"):
            act = ops.adam_active_flags(p)
            tch = on.to(torch.uint8).contiguous()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for it in range(6):
                if kind == "touched":
                    tch.copy_(on.to(torch.uint8))
                torch.cuda.synchronize()
                e0.record()
                ops.adam_active_(p, g, m, v, act, it + 1, 1e-3, touched=tch if kind == "touched" else None)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res.append(min(ts[1:]))
        print(f"{name:28s} moving {frac:4.2f}: active {res[0]:7.1f} us  touched {res[1]:7.1f} us")
        del p, m, v, g
