"""Do the next batch's sort and the current step's gradient pull overlap when they run on two streams (dev probe for
DESIGN 4.10 item 3)?  cfg-2 shapes; serial = both on one stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402

dev = "cuda:0"
n, L, C = 262144, 3, 8
torch.manual_seed(0)
feats = [(torch.randn(1, C, s, s, s, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
         for s in (32, 64, 128)]
meta = ops.GridMeta.from_bound([[-1.0, 1.0]] * 3)
x = (torch.rand(n, 3, device=dev) * 2 - 1)
x2 = (torch.rand(n, 3, device=dev) * 2 - 1)
sb = ops.SortedBatch(n, dev).sort(x, meta)
sb2 = ops.SortedBatch(n, dev)
df = torch.randn(n, L * C, device=dev)
grads = [torch.empty_like(f) for f in feats]
side = torch.cuda.Stream()


def run(overlap, iters=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        if overlap:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                sb2.sort(x2, meta)
            ops.grad_pull_raw(feats, meta, sb, df, grads, overwrite=True)
            torch.cuda.current_stream().wait_stream(side)
        else:
            sb2.sort(x2, meta)
            ops.grad_pull_raw(feats, meta, sb, df, grads, overwrite=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for _ in range(2):
    run(False, 5), run(True, 5)
print(f"sort + pull on one stream: {run(False):.1f} us; sort on a second stream: {run(True):.1f} us")
