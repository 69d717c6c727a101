#!/usr/bin/env python3
"""dev: the fused forward at cfg-2 (sorted batch) under the MISO_TUNE ablations (2 = no gather, 4 = gather only) and
grid caps (MISO_FWD_BLOCKS): how long the matrix phase and the gather phase take with 1 / 2 / 3 / 4 waves per SIMD."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import ops
from tools.quick_bench import timeit
dev = "cuda:0"; n = 262144; L, C, H = 3, 8, 64
torch.manual_seed(0)
feats = [(torch.randn(1, C, s, s, s, device=dev) * 1e-2).contiguous(memory_format=torch.channels_last_3d) for s in (32, 64, 128)]
meta = ops.GridMeta.from_bound([[-1.0, 1.0]] * 3)
lin = [torch.nn.Linear(L * C, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
pack = ops.DecoderPack([l.weight.data.to(dev) for l in lin], [l.bias.data.to(dev) for l in lin])
x = (torch.rand(n, 3, generator=torch.Generator().manual_seed(1234)) * 2 - 1).to(dev)
sb = ops.SortedBatch(n, dev).sort(x, meta)
_, mask = ops.sdf_fwd_raw(x, feats, meta, pack, True, sorted_batch=sb)
t = timeit(lambda: ops.sdf_fwd_raw(x, feats, meta, pack, True, mask=mask, sorted_batch=sb))
print(f"TUNE={os.environ.get('MISO_TUNE','0')} BLOCKS={os.environ.get('MISO_FWD_BLOCKS','512')}: {t:.1f} us")
