"""dev: sdf_train_kernel / pull / whole-step durations of the cfg-2 headline workload for the library MISO_HIP_LIB points at
(default: the in-tree one).  tools/train_ab.sh runs it over every library under _ab/."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from miso_amd import ops  # noqa: E402


def main():
    dev = "cuda:0"
    step, _ = bench.build_workload(dev, 0)
    for _ in range(300):
        step.run()
    torch.cuda.synchronize()
    feats, meta, pack, sb = step.features, step.meta, step.pack, step.sorted
    L = len(feats)
    t_train = bench.time_kernel(lambda: ops.sdf_train_raw(feats, meta, pack, sb, step.aux, step.loss_slots, step.grads,
                                                          "L1", 1.0, 0.0, 0.0))
    sb2 = ops.SortedBatch(bench.N_POINTS, dev, tiles=step.tiles).sort(step.x, meta)
    ws = sb2.bwd_workspace(bench.N_POINTS * L * bench.C)
    t_pull = bench.time_kernel(lambda: ops.grad_pull_raw(feats, meta, sb2, ws, step.grads, overwrite=True))
    t_step = bench.time_kernel(step.run, iters=200)
    mask = torch.empty(((bench.N_POINTS + 63) // 64) * 64 * ops.sdf_mask_words(pack), device=dev, dtype=torch.int32)
    t_fwd = bench.time_kernel(lambda: ops.sdf_fwd_loss_raw(feats, meta, pack, sb2, step.aux, mask, step.gpred,
                                                           step.loss_slots, "L1", 1.0, 0.0, 0.0, sdf_out=None))
    t_bwd = bench.time_kernel(lambda: ops.sdf_bwd_raw(step.x, feats, meta, pack, step.gpred, mask, False, [True] * L,
                                                      step.grads, sorted_batch=sb2, overwrite=True, gsdf_sorted=True))
    print(f"{os.environ.get('MISO_HIP_LIB', 'in-tree'):40s} train {t_train - t_pull:6.1f} us  pull {t_pull:6.1f}  step {t_step:6.1f}"
          f"  fwd {t_fwd:6.1f}  bwd(mfma pass) {t_bwd - t_pull:6.1f}")


if __name__ == "__main__":
    main()
