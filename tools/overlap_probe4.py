"""dev probe: the NEXT batch's sort (small workgroups, MISO_SORT_SMALL=1) on a second stream under the current batch's
training kernel -- stream launches only, for a kernel trace (tools/ovl_trace.py prints a window of it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402

dev = "cuda:0"
n, L, C, H = 262144, 3, 8, 64
torch.manual_seed(0)
feats = [(torch.randn(1, C, s, s, s, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
         for s in (32, 64, 128)]
meta = ops.GridMeta.from_bound([[-1.0, 1.0]] * 3)
lin = [torch.nn.Linear(L * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
pack = ops.DecoderPack([m.weight.detach().to(dev) for m in lin], [m.bias.detach().to(dev) for m in lin])
xs = [(torch.rand(n, 3, device=dev) * 2 - 1) for _ in range(2)]
sbs = [ops.SortedBatch(n, dev) for _ in range(2)]
sbs[0].sort(xs[0], meta)
aux = torch.cat((torch.randn(n, 1, device=dev) * 0.1, torch.ones(n, 3, device=dev)), dim=1).contiguous()
slots = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=dev)
grads = [torch.empty_like(f) for f in feats]
side = torch.cuda.Stream()
order = os.environ.get("ORDER", "sort_first")


def body(p):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    if order == "sort_first":
        with torch.cuda.stream(side):
            sbs[1 - p].sort(xs[1 - p], meta)
        ops.sdf_train_raw(feats, meta, pack, sbs[p], aux, slots, grads, "L1", 1.0, 0.0, 0.0)
    else:
        ops.sdf_train_raw(feats, meta, pack, sbs[p], aux, slots, grads, "L1", 1.0, 0.0, 0.0)
        with torch.cuda.stream(side):
            sbs[1 - p].sort(xs[1 - p], meta)
    cur.wait_stream(side)


for i in range(300):
    body(i & 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(200):
    body(i & 1)
e1.record()
torch.cuda.synchronize()
print(f"order={order}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per step")
