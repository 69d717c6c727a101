"""dev probe: the headline MappingStep as a graph replay vs as plain stream launches (cfg-2), same protocol as bench.py."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from miso_amd.step import MappingStep  # noqa: E402

dev = torch.device("cuda:0")
eager, (feats, ws, bs, x, targ) = bench.build_workload(dev, 0)
assert not eager._use_graph
step = MappingStep(eager.features, eager.meta, eager.pack, bench.N_POINTS, loss_type="L1", weight_sdf=1.0, weight_fs=0.0,
                   keep_sdf=False, use_graph=True)
step.set_batch(x.to(dev), targ.to(dev))


def run(s, k=200):
    for _ in range(20):
        s.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        s.run()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6, t_host / k * 1e6


for k in (200, 200, 200, 200, 20, 20, 3000):
    a, b = run(step, k), run(eager, k)
    print(f"{k} steps: graph {a[0]:.1f} us/step (host {a[1]:.1f}); eager {b[0]:.1f} us/step (host enqueue {b[1]:.1f})")
