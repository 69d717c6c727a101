"""dev: per-step host enqueue time of the stream-launched headline step, to find the slow stretch seen in
tools/graph_vs_eager.py (third run of 200)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
step, _ = bench.build_workload(dev, 0)
step.AHEAD_EVERY = int(os.environ.get("AHEAD", "8"))
ts = []
import gc
if os.environ.get("GCFREEZE"):
    gc.collect()
    gc.freeze()
torch.cuda.synchronize()
t00 = time.perf_counter()
for i in range(1500):
    t0 = time.perf_counter()
    step.run()
    ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print(f"total {(time.perf_counter() - t00) / 1500 * 1e6:.1f} us/step")
slow = [(i, round(t * 1e6)) for i, t in enumerate(ts) if t > 400e-6]
print("steps with host time > 400 us:", len(slow), slow[:40])
import statistics
print("median host", statistics.median(ts) * 1e6)
