"""dev: miso_adam_touched on a 145 M-float level (the Newer College fine level) for synthetic flag patterns -- what
bounds it: the flag scan, the number of stepped chunks, or where they lie?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops  # noqa: E402

dev = "cuda:0"
n = 600 * 600 * 100 * 4
p = torch.zeros(n, device=dev)
g, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
CH = ops._lib.ADAM_CHUNK
nch = (n + CH - 1) // CH
act = torch.zeros(nch, device=dev, dtype=torch.uint8)
tch = torch.zeros(nch, device=dev, dtype=torch.uint8)
gen = torch.Generator(device=dev).manual_seed(1)


def timed(pattern, iters=30):
    for _ in range(5):
        act.zero_(); tch.copy_(pattern)
        ops.adam_active_(p, g, m, v, act, 1, 1e-3, zero_grad=True, touched=tch)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        act.zero_(); tch.copy_(pattern)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.adam_active_(p, g, m, v, act, 1, 1e-3, zero_grad=True, touched=tch)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


pat = torch.zeros(nch, device=dev, dtype=torch.uint8)
print(f"chunk {CH} floats, {nch} flags; no chunk set: {timed(pat):.1f} us")
for k in (1000, 5000, 25000, 100000):
    pat.zero_()
    pat[torch.randint(0, nch, (k,), device=dev, generator=gen)] = 1
    print(f"{k} chunks at random: {timed(pat):.1f} us")
    pat.zero_()
    pat[nch // 3: nch // 3 + k] = 1
    print(f"{k} consecutive chunks: {timed(pat):.1f} us")
# the bench's pattern: samples around the sensor
x = (torch.rand(6144, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1) * torch.tensor([25.0, 25.0, 4.0]) + torch.tensor([5.0, -8.0, 2.0])
ix = ((x + torch.tensor([60.0, 60.0, 5.0])) / 0.2).long()
pat.zero_()
for dz in (0, 1):
    for dy in (0, 1):
        for dx in (0, 1):
            off = (((ix[:, 2] + dz) * 600 + ix[:, 1] + dy) * 600 + ix[:, 0] + dx) * 4
            pat[(off // CH).to(dev)] = 1
print(f"the 6 144-sample batch of the trainer bench ({int(pat.sum())} chunks): {timed(pat):.1f} us")
