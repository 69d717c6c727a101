"""dev: run the headline step N times as graph replays or as stream launches (for a kernel trace of either)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from miso_amd.step import MappingStep  # noqa: E402

mode, k = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
step, (feats, ws, bs, x, targ) = bench.build_workload(dev, 0)
if mode == "eager":
    step = MappingStep(step.features, step.meta, step.pack, bench.N_POINTS, loss_type="L1", weight_sdf=1.0, weight_fs=0.0,
                       keep_sdf=False, use_graph=False)
    step.set_batch(x.to(dev), targ.to(dev))
for _ in range(k):
    step.run()
torch.cuda.synchronize()
