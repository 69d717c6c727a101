"""The headline step alone (dev): us per MappingStep.run() at cfg-2 after the settle phase, best of five loops of 200."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
step, _ = bench.build_workload(dev, 0)
for _ in range(400):
    step.run()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(200):
        step.run()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
print(f"step {best:.1f} us")
