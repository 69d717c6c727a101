"""cfg-4 alone (dev / profiling): 8 ScanNet-shaped submaps, 28 pairs, the fused alignment loop at levels 0 and 1."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
atlas = bench.scannet_atlas(dev, 8)
atlas.precompute_coordinates_for_alignment()
print(json.dumps(bench.align_cfg4(dev, atlas, iters=int(os.environ.get("ITERS", 20))), indent=1))
