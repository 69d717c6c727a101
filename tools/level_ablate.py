"""dev: the fused forward / training kernels at cfg-2 with levels switched off (ignore_level): what each level's corner
gathers cost.  Ignored levels contribute zero features; the decoder runs unchanged."""
import dataclasses
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
    feats, pack = step.features, step.pack
    sb = ops.SortedBatch(bench.N_POINTS, dev, tiles=step.tiles).sort(step.x, step.meta)
    mask = torch.empty(((bench.N_POINTS + 63) // 64) * 64 * ops.sdf_mask_words(pack), device=dev, dtype=torch.int32)
    for ig in ([0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 1, 1]):
        meta = dataclasses.replace(step.meta, ignore_mask=sum(1 << l for l, i in enumerate(ig) if i))
        t_fwd = bench.time_kernel(lambda: ops.sdf_fwd_loss_raw(feats, meta, pack, sb, step.aux, mask, step.gpred,
                                                               step.loss_slots, "L1", 1.0, 0.0, 0.0, sdf_out=None))
        grads = [None if i else g for i, g in zip(ig, step.grads)]
        t_pull = 0.0
        if not all(ig):
            ws = sb.bwd_workspace(bench.N_POINTS * len(feats) * bench.C)
            t_pull = bench.time_kernel(lambda: ops.grad_pull_raw(feats, meta, sb, ws, grads, overwrite=True))
        try:
            t_tr = bench.time_kernel(lambda: ops.sdf_train_raw(feats, meta, pack, step.sorted, step.aux, step.loss_slots,
                                                               grads, "L1", 1.0, 0.0, 0.0))
        except RuntimeError:      # (the one-launch form wants a gradient for every live level)
            t_tr = float("nan")
        print(f"ignored (coarse, mid, fine) = {ig}: fwd {t_fwd:6.1f} us   train+pull {t_tr:6.1f}   pull {t_pull:6.1f}"
              f"   train {t_tr - t_pull:6.1f}")


if __name__ == "__main__":
    main()
