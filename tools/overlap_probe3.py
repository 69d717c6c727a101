"""dev probe (cfg-3 shapes): does the forward of one batch overlap with the atomic-bound backward of another when the
two run on two streams?  If the pair takes ~max instead of ~sum, a fused forward + backward (atomics included) kernel
would hide the fine level's atomics behind matrix work."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402

dev = "cuda:0"
n, C, H = 540000, 4, 64
torch.manual_seed(0)
sizes = [(40, 20, 40), (200, 100, 200)]
mk = lambda: [(torch.randn(1, C, z, y, x, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d)
              for (x, y, z) in sizes]
featsA, featsB = mk(), mk()
meta = ops.GridMeta.from_bound([[-10.0, 10.0], [-5.0, 5.0], [-10.0, 10.0]])
lin = [torch.nn.Linear(2 * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
pack = ops.DecoderPack([m.weight.detach().to(dev) for m in lin], [m.bias.detach().to(dev) for m in lin])
x = ((torch.rand(n, 3) * 2 - 1) * torch.tensor([6.0, 2.5, 6.0])).to(dev)
sb = ops.SortedBatch(n, dev).sort(x, meta)
aux = torch.cat((torch.randn(n, 1, device=dev) * 0.1, torch.ones(n, 3, device=dev)), dim=1).contiguous()
mw = ops.sdf_mask_words(pack)
maskA = torch.empty(((n + 63) // 64) * 64 * mw, device=dev, dtype=torch.int32)
maskB = torch.empty_like(maskA)
gpA, gpB = torch.empty(n, 1, device=dev), torch.empty(n, 1, device=dev)
slots = torch.zeros(ops._lib.LOSS_SLOTS, 2, device=dev)
slots2 = torch.zeros_like(slots)
gradsB = [torch.zeros_like(f) for f in featsB]
ops.sdf_fwd_loss_raw(featsB, meta, pack, sb, aux, maskB, gpB, slots2, "L1", 1.0, 0.0, 0.0)
side = torch.cuda.Stream()


def fwd():
    ops.sdf_fwd_loss_raw(featsA, meta, pack, sb, aux, maskA, gpA, slots, "L1", 1.0, 0.0, 0.0)


def bwd():
    ops.sdf_bwd_raw(x, featsB, meta, pack, gpB, maskB, False, [True, True], gradsB, sorted_batch=sb, overwrite=True,
                    gsdf_sorted=True)


def run(mode, iters=40):
    cur = torch.cuda.current_stream()
    for _ in range(5):
        fwd(); bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        if mode == "fwd":
            fwd()
        elif mode == "bwd":
            bwd()
        elif mode == "serial":
            fwd(); bwd()
        else:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                if mode == "bwd_first":
                    bwd()
                else:
                    fwd()
            if mode == "bwd_first":
                fwd()
            else:
                bwd()
            cur.wait_stream(side)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for mode in ("fwd", "bwd", "serial", "fwd_first", "bwd_first"):
    print(f"{mode}: {run(mode):.1f} us")
