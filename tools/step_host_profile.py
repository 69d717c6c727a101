"""dev: cProfile of the host side of the stream-launched headline step (where do its 35 us go?)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
step, _ = bench.build_workload(dev, 0)
for _ in range(50):
    step.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    step.run()
print(f"host per step, 40 steps into an empty queue: {(time.perf_counter() - t0) / 40 * 1e6:.1f} us")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    step.run()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
