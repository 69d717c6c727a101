"""TEST-ONLY: stand-ins for the HIP operators built from the CPU oracle, so that the
host-side logic of miso_amd.grid_opt (trainers, losses, alignment loops, sharding) can
be exercised on a machine without a GPU.  Installed by the ``oracle_ops`` fixture via
monkeypatch; the product package never imports this module."""
import torch

from oracle import ref_torch as R


def _bound(meta):
    return torch.tensor([[meta.bound_min[a], meta.bound_max[a]] for a in range(3)], dtype=torch.float32)


def encode(x, features, meta):
    from miso_amd import _lib
    ignore = [(meta.ignore_mask >> l) & 1 == 1 for l in range(len(features))]
    if meta.flags & _lib.F_COORDS_NORMALIZED:
        outs = []
        for l, f in enumerate(features):
            v = R.trilinear_gather(f, x, bool(meta.flags & _lib.F_ALIGN_CORNERS),
                                   "border" if meta.flags & _lib.F_PAD_BORDER else "zeros")
            outs.append(torch.zeros_like(v) if ignore[l] else v)
        return torch.cat(outs, dim=1)
    return R.encode_gather(list(features), _bound(meta).to(x), x, ignore)


def grid_sample_3d(input, grid, padding_mode="zeros", align_corners=True):
    _, do, ho, wo, _ = grid.shape
    out = R.trilinear_gather(input, grid.reshape(-1, 3), align_corners, padding_mode)
    return out.transpose(0, 1).reshape(1, input.shape[1], do, ho, wo)


def mapping_loss(pred, target, valid, sign, weight, loss_type="L1", weight_sdf=1.0, weight_fs=0.0,
                 trunc_dist=0.0):
    a = weight_sdf * R.miso_loss_regression(pred, target, valid, weight, loss_type)
    b = weight_fs * R.miso_loss_free_space(pred, target, sign, trunc_dist) if weight_fs > 0 \
        else torch.zeros((), dtype=pred.dtype)
    return torch.stack((a, b))


def install(monkeypatch):
    from miso_amd import ops
    monkeypatch.setattr(ops, "encode", encode)
    monkeypatch.setattr(ops, "grid_sample_3d", grid_sample_3d)
    monkeypatch.setattr(ops, "mapping_loss", mapping_loss)
    monkeypatch.setattr(ops, "sdf_fused_supported", lambda *a, **k: False)
