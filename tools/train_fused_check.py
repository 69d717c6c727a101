"""dev: the one-launch training kernel (sdf_train_kernel) against the two-launch form on the cfg-2 step: same loss, same
gradients (one sorted order), and the step time of both as graph replays."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from miso_amd import ops  # noqa: E402
from tools.quick_bench import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    step, _ = bench.build_workload(dev, 0)
    assert step._fused_train()
    # same sorted batch for both forms: sort once, call the two paths by hand
    sb = ops.SortedBatch(step.n, dev, tiles=step.tiles).sort(step.x, step.meta)      # with perm[]: the two-launch form needs it
    L = len(step.features)
    mw = ops.sdf_mask_words(step.pack)
    mask = torch.empty(((step.n + 63) // 64) * 64 * mw, device=dev, dtype=torch.int32)
    g2 = [torch.empty_like(f) for f in step.features]
    slots2 = torch.zeros_like(step.loss_slots)
    ops.sdf_fwd_loss_raw(step.features, step.meta, step.pack, sb, step.aux, mask, step.gpred, slots2, "L1", 1.0, 0.0, 0.0)
    ops.sdf_bwd_raw(step.x, step.features, step.meta, step.pack, step.gpred, mask, False, [True] * L, g2,
                    sorted_batch=sb, overwrite=True, gsdf_sorted=True)
    g1 = [torch.empty_like(f) for f in step.features]
    slots1 = torch.zeros_like(step.loss_slots)
    sdf1 = torch.empty(step.n, 1, device=dev)
    ops.sdf_train_raw(step.features, step.meta, step.pack, sb, step.aux, slots1, g1, "L1", 1.0, 0.0, 0.0, sdf_out=sdf1)
    torch.cuda.synchronize()
    print("loss", slots1.sum(0).tolist(), slots2.sum(0).tolist(), "slots equal:", torch.equal(slots1, slots2))
    for l in range(L):
        d = (g1[l] - g2[l]).abs().max().item()
        print(f"level {l}: max |d grad| {d:.3e} of {g2[l].abs().max().item():.3e}, bitwise {torch.equal(g1[l], g2[l])}")
    sdf2, _ = ops.sdf_fwd_raw(step.x, step.features, step.meta, step.pack, False)
    print("sdf max diff", (sdf1 - sdf2).abs().max().item())
    for name, env in (("fused", None), ("two launches", "1")):
        if env:
            os.environ["MISO_NO_FUSED_TRAIN"] = env
        st, _ = bench.build_workload(dev, 0)
        for _ in range(5):
            st.run()
        t = timeit(st.run, iters=200, warm=20)
        print(f"{name}: {t:.1f} us per step -> {step.n / t * 1e-3:.3f} G point-samples/s")
        os.environ.pop("MISO_NO_FUSED_TRAIN", None)


if __name__ == "__main__":
    main()
