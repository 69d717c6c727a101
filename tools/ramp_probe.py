"""dev: duration of consecutive blocks of 5 headline steps (a) from a cold device, (b) after 40 ms of steps followed by a
synchronize and an idle gap of X ms -- how long does the device keep its clocks?"""
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
NB = 16
evs = [torch.cuda.Event(enable_timing=True) for _ in range(NB + 1)]


def blocks():
    evs[0].record()
    for b in range(NB):
        for _ in range(5):
            step.run()
        evs[b + 1].record()
    torch.cuda.synchronize()
    return " ".join(f"{evs[b].elapsed_time(evs[b + 1]) / 5 * 1e3:.0f}" for b in range(NB))


print("cold:", blocks())
for idle_ms in (0.0, 0.1, 0.5, 1.0, 2.0, 5.0, 20.0, 100.0):
    for _ in range(250):
        step.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < idle_ms * 1e-3:
        pass
    print(f"warm, then idle {idle_ms} ms:", blocks())
