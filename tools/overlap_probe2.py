"""dev probe: does the NEXT batch's sort hide behind the current batch's sdf_train_kernel + pull when it runs on a
second stream?  cfg-2 shapes.  serial = one stream."""
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


def body(p, overlap):
    cur = torch.cuda.current_stream()
    if overlap:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            sbs[1 - p].sort(xs[1 - p], meta)
        ops.sdf_train_raw(feats, meta, pack, sbs[p], aux, slots, grads, "L1", 1.0, 0.0, 0.0)
        cur.wait_stream(side)
    else:
        sbs[1 - p].sort(xs[1 - p], meta)
        ops.sdf_train_raw(feats, meta, pack, sbs[p], aux, slots, grads, "L1", 1.0, 0.0, 0.0)


def run(overlap, iters=100, graph=True):
    if graph:
        gs = []
        for p in (0, 1):
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body(p, overlap)        # warm
                g.capture_begin()
                body(p, overlap)
                g.capture_end()
            torch.cuda.current_stream().wait_stream(s)
            gs.append(g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        if graph:
            gs[i & 1].replay()
        else:
            body(i & 1, overlap)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for graph in (False, True):
    for _ in range(2):
        a, b = run(False, graph=graph), run(True, graph=graph)
    print(f"graph={graph}: sort + train + pull on one stream {a:.1f} us; sort of the next batch on a second stream {b:.1f} us")
g0 = [g.clone() for g in grads]
body(0, False); body(1, False)
torch.cuda.synchronize()
print("grads finite", all(torch.isfinite(g).all().item() for g in grads))
