"""dev: us per headline step in blocks of 250 steps from the very first step of a process on a box whose GPU has been idle
for a while -- how long until the sustained rate?"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
step, _ = bench.build_workload(dev, 0)
gc.collect()
gc.freeze()
gc.disable()
out = []
t_start = time.perf_counter()
for b in range(int(os.environ.get("BLOCKS", "40"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(250):
        step.run()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t_start, (time.perf_counter() - t0) / 250 * 1e6))
print(" ".join(f"{t:.2f}s:{u:.1f}" for t, u in out))
