"""The library's entry points as ``torch.library`` custom operators (namespace ``miso``).

SURVEY 8(b): the reference reaches its native code through one registered operator (``gridsample_grad2.grad2_3d``,
third_party/cuda_gridsample_grad2/gridsample_cuda.cpp:53-56) and two ATen ones (``aten::grid_sampler_3d`` and
``aten::grid_sampler_3d_backward``, cuda_gridsample.py:84,102) behind ``autograd.Function`` wrappers.  The product path
of this package does the same with ``autograd.Function`` objects over the C-ABI (``miso_amd.ops``); this module exposes
the SAME calls as dispatcher-visible operators, so that ``torch.compile`` / ``torch.export`` trace through them (fake
kernels give the shapes), ``torch.library.opcheck`` can test them, and a caller that holds only tensors -- no
``GridMeta`` / ``DecoderPack`` objects -- can call them:

    miso::encode_fwd           multires trilinear encode                     (A2, A5: grid_modules.py:72-95, utils.py:143-164)
    miso::encode_bwd           its first backward                            (A3: cuda_gridsample.py:99-113)
    miso::encode_bwd2          its second backward                           (A4: gridsample_cuda.cu:212-533)
    miso::encode_decode_fwd    encode + frozen MLP decoder                   (A7: grid_net.py:306-325)
    miso::decode_bwd           backward of the above (decoder frozen)        (A3 + A6)
    miso::pair_latent_fwd_bwd  alignment residual + pose cotangent sums      (A13e: align/miso.py:116-211)
    miso::lm_normal_eq         tracking normal equations                     (A14: slam/tracker.py:148-212)
    miso::adam_dense           in-place Adam step on one dense tensor        (A11: trainer.py:410-480)

A submap's bound travels as six floats (min x y z, max x y z), its level mask and sampling flags as ints (``GridMeta``).
Absent optional results (a gradient nobody asked for) come back as empty tensors: operator schemas have no
``Optional[Tensor]`` results.  ``encode_fwd`` differentiates to second order through ``encode_bwd`` -> ``encode_bwd2``
(``create_graph=True``: the eikonal / smoothness losses of loss_isdf.py:367-377); ``encode_decode_fwd`` to first order.
Device: the HIP device only -- there is no CPU kernel behind any of them (the CPU restatement lives in ``oracle/`` and is
test infrastructure).
"""
from typing import List, Sequence, Tuple

import torch

from . import ops

_LIB = "miso"


def _meta(bound: Sequence[float], ignore_mask: int, flags: int) -> ops.GridMeta:
    assert len(bound) == 6, "bound = (min x, min y, min z, max x, max y, max z)"
    return ops.GridMeta(tuple(float(v) for v in bound[:3]), tuple(float(v) for v in bound[3:]), int(ignore_mask), int(flags))


def meta_args(meta: ops.GridMeta) -> Tuple[List[float], int, int]:
    """GridMeta -> the (bound, ignore_mask, flags) arguments of the operators."""
    return list(meta.bound_min) + list(meta.bound_max), int(meta.ignore_mask), int(meta.flags)


def _or_empty(t, like: torch.Tensor) -> torch.Tensor:
    return t if t is not None else like.new_empty(0)


def _or_none(t):
    return t if t is not None and t.numel() else None


# --------------------------------------------------------------------------- #
# encode, to second order
# --------------------------------------------------------------------------- #
@torch.library.custom_op(f"{_LIB}::encode_fwd", mutates_args=(), device_types="cuda")
def encode_fwd(x: torch.Tensor, features: Sequence[torch.Tensor], bound: Sequence[float], ignore_mask: int,
               flags: int) -> torch.Tensor:
    return ops.encode_fwd_raw(x, list(features), _meta(bound, ignore_mask, flags))


@encode_fwd.register_fake
def _(x, features, bound, ignore_mask, flags):
    return x.new_empty((x.shape[0], sum(f.shape[1] for f in features)))


@torch.library.custom_op(f"{_LIB}::encode_bwd", mutates_args=(), device_types="cuda")
def encode_bwd(gout: torch.Tensor, x: torch.Tensor, features: Sequence[torch.Tensor], bound: Sequence[float],
               ignore_mask: int, flags: int, need_x: bool, need_f: Sequence[bool]) -> List[torch.Tensor]:
    """-> [grad x (N,3) or empty, grad of every level (its layout) or empty]"""
    gx, grads = ops.encode_bwd_raw(x, list(features), _meta(bound, ignore_mask, flags), gout, need_x, list(need_f))
    return [_or_empty(gx, x)] + [_or_empty(g, x) for g in grads]


@encode_bwd.register_fake
def _(gout, x, features, bound, ignore_mask, flags, need_x, need_f):
    return [x.new_empty((x.shape[0], 3) if need_x else 0)] + \
           [torch.empty_like(f) if nf else x.new_empty(0) for f, nf in zip(features, need_f)]


@torch.library.custom_op(f"{_LIB}::encode_bwd2", mutates_args=(), device_types="cuda")
def encode_bwd2(gout: torch.Tensor, x: torch.Tensor, features: Sequence[torch.Tensor], bound: Sequence[float],
                ignore_mask: int, flags: int, ggx: torch.Tensor, ggf: Sequence[torch.Tensor], need_x: bool,
                need_f: Sequence[bool]) -> List[torch.Tensor]:
    """Cotangents of encode_bwd's results (ggx: of grad x, ggf: of the level gradients; empty = none) ->
    [d / d gout (N,F), d / d x (N,3) or empty, d / d level or empty]"""
    gg_out, g_x, g_f = ops.encode_bwd2_raw(x, list(features), _meta(bound, ignore_mask, flags), gout, _or_none(ggx),
                                           [_or_none(t) for t in ggf], need_x, list(need_f))
    return [gg_out, _or_empty(g_x, x)] + [_or_empty(g, x) for g in g_f]


@encode_bwd2.register_fake
def _(gout, x, features, bound, ignore_mask, flags, ggx, ggf, need_x, need_f):
    want = [bool(nf) and ggx.numel() > 0 for nf in need_f]
    return [x.new_empty((x.shape[0], sum(f.shape[1] for f in features))), x.new_empty((x.shape[0], 3) if need_x else 0)] + \
           [torch.empty_like(f) if w else x.new_empty(0) for f, w in zip(features, want)]


def _encode_setup(ctx, inputs, output):
    x, features, bound, ignore_mask, flags = inputs
    ctx.save_for_backward(x, *features)
    ctx.args = (bound, ignore_mask, flags)


def _encode_backward(ctx, gout):
    x, *features = ctx.saved_tensors
    need_x = ctx.needs_input_grad[0]
    need_f = [f.requires_grad for f in features]
    res = encode_bwd(gout.contiguous(), x, features, *ctx.args, need_x, need_f)
    return (_or_none(res[0]) if need_x else None, [_or_none(g) if nf else None for g, nf in zip(res[1:], need_f)],
            None, None, None)


encode_fwd.register_autograd(_encode_backward, setup_context=_encode_setup)


def _encode_bwd_setup(ctx, inputs, output):
    gout, x, features, bound, ignore_mask, flags, need_x, need_f = inputs
    ctx.save_for_backward(gout, x, *features)
    ctx.args = (bound, ignore_mask, flags)
    ctx.n_levels = len(features)


def _encode_bwd_backward(ctx, grads_out):
    gout, x, *features = ctx.saved_tensors
    ggx, ggf = grads_out[0], list(grads_out[1:])
    empty = x.new_empty(0)
    need_gout, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    need_f = [f.requires_grad for f in features]
    ggx_t = ggx if ggx is not None and ggx.numel() else empty
    ggf_t = [t if t is not None and t.numel() else empty for t in ggf]
    if not ggx_t.numel() and not any(t.numel() for t in ggf_t):
        return (None,) * 8
    res = encode_bwd2(gout, x, features, *ctx.args, ggx_t, ggf_t, need_x, need_f)
    return (res[0] if need_gout else None, _or_none(res[1]) if need_x else None,
            [_or_none(g) if nf else None for g, nf in zip(res[2:], need_f)], None, None, None, None, None)


encode_bwd.register_autograd(_encode_bwd_backward, setup_context=_encode_bwd_setup)


# --------------------------------------------------------------------------- #
# encode + frozen decoder
# --------------------------------------------------------------------------- #
import collections

_PACKS_MAX = 16
_packs = collections.OrderedDict()      # least recently used first


def _pack(weights, biases) -> ops.DecoderPack:
    """One DecoderPack per set of weight tensors (the pack re-permutes itself when a weight's version changes).
    Keyed by storage AND layout -- two views of one buffer (same base pointer, another shape or stride) are different
    decoders -- and by device; a small LRU: the oldest entry goes when the 17th arrives, so discarded models' weights
    are not kept alive indefinitely and nothing is dropped in the middle of a forward / backward pair."""
    def k(t):
        return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.device.index if t.is_cuda else -1)
    key = tuple(k(w) for w in weights) + tuple(k(b) for b in biases)
    p = _packs.get(key)
    if p is not None:
        _packs.move_to_end(key)
        return p
    while len(_packs) >= _PACKS_MAX:
        _packs.popitem(last=False)
    p = _packs[key] = ops.DecoderPack(list(weights), [b if b.numel() else None for b in biases])
    return p


@torch.library.custom_op(f"{_LIB}::encode_decode_fwd", mutates_args=(), device_types="cuda")
def encode_decode_fwd(x: torch.Tensor, features: Sequence[torch.Tensor], weights: Sequence[torch.Tensor],
                      biases: Sequence[torch.Tensor], bound: Sequence[float], ignore_mask: int, flags: int,
                      want_mask: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (sdf (N,1), ReLU sign bits for decode_bwd (int32 words; empty without want_mask)).  biases: one tensor per
    linear layer, empty = no bias (MLPNet(bias=False), modules.py:11-40)."""
    sdf, mask = ops.sdf_fwd_raw(x, list(features), _meta(bound, ignore_mask, flags), _pack(weights, biases), want_mask)
    return sdf, mask if mask is not None else torch.empty(0, device=x.device, dtype=torch.int32)


@encode_decode_fwd.register_fake
def _(x, features, weights, biases, bound, ignore_mask, flags, want_mask):
    n = x.shape[0]
    words = (len(weights) - 1) * weights[0].shape[0] // 32        # H bits per hidden activation layer and point
    return x.new_empty((n, 1)), torch.empty(((n + 63) // 64) * 64 * words if want_mask else 0, device=x.device,
                                            dtype=torch.int32)


@torch.library.custom_op(f"{_LIB}::decode_bwd", mutates_args=(), device_types="cuda")
def decode_bwd(gsdf: torch.Tensor, x: torch.Tensor, features: Sequence[torch.Tensor], weights: Sequence[torch.Tensor],
               biases: Sequence[torch.Tensor], mask: torch.Tensor, bound: Sequence[float], ignore_mask: int, flags: int,
               need_x: bool, need_f: Sequence[bool]) -> List[torch.Tensor]:
    """-> [grad x (N,3) or empty, grad of every level or empty] (the decoder is frozen: no weight gradients)"""
    gx, grads = ops.sdf_bwd_raw(x, list(features), _meta(bound, ignore_mask, flags), _pack(weights, biases), gsdf, mask,
                                need_x, list(need_f))
    return [_or_empty(gx, x)] + [_or_empty(g, x) for g in grads]


@decode_bwd.register_fake
def _(gsdf, x, features, weights, biases, mask, bound, ignore_mask, flags, need_x, need_f):
    return [x.new_empty((x.shape[0], 3) if need_x else 0)] + \
           [torch.empty_like(f) if nf else x.new_empty(0) for f, nf in zip(features, need_f)]


def _sdf_setup(ctx, inputs, output):
    x, features, weights, biases, bound, ignore_mask, flags, want_mask = inputs
    if not want_mask:
        raise RuntimeError("miso::encode_decode_fwd: want_mask=True is needed to differentiate it")
    ctx.save_for_backward(x, output[1], *features, *weights, *biases)
    ctx.counts = (len(features), len(weights))
    ctx.args = (bound, ignore_mask, flags)


def _sdf_backward(ctx, gsdf, _gmask):
    x, mask, *rest = ctx.saved_tensors
    nf_, nw = ctx.counts
    features, weights, biases = rest[:nf_], rest[nf_:nf_ + nw], rest[nf_ + nw:]
    need_x = ctx.needs_input_grad[0]
    need_f = [f.requires_grad for f in features]
    res = decode_bwd(gsdf.contiguous(), x, features, weights, biases, mask, *ctx.args, need_x, need_f)
    return (_or_none(res[0]) if need_x else None, [_or_none(g) if nf else None for g, nf in zip(res[1:], need_f)],
            [None] * nw, [None] * nw, None, None, None, None)


encode_decode_fwd.register_autograd(_sdf_backward, setup_context=_sdf_setup)


# --------------------------------------------------------------------------- #
# alignment, tracking, optimiser
# --------------------------------------------------------------------------- #
@torch.library.custom_op(f"{_LIB}::pair_latent_fwd_bwd", mutates_args=(), device_types="cuda")
def pair_latent_fwd_bwd(pose: torch.Tensor, coords_src: torch.Tensor, feats_src: torch.Tensor,
                        feats_dst: Sequence[torch.Tensor], bound: Sequence[float], ignore_mask: int, flags: int,
                        loss_type: int) -> torch.Tensor:
    """pose = (R_src 9, t_src 3, R_dst 9, t_dst 3) -> 24 fp64 sums of ``miso_pair_latent`` (include/miso_hip.h):
    [0] loss numerator, [1] in-bound count, [2:5] sum of g_i, [5:14] d / d R_dst, [14:23] d / d R_src."""
    import ctypes as C
    from . import _lib
    ops._require_hip(pose, coords_src, feats_src, *feats_dst)
    n = coords_src.shape[0]
    coords_src = coords_src.contiguous()
    feats_src = ops._rows(feats_src)
    n_ch = ops._feature_dim(feats_dst)
    out = torch.empty(24, device=coords_src.device, dtype=torch.float64)
    g = ops._fill_grid(list(feats_dst), _meta(bound, ignore_mask, flags))
    _lib.check(_lib.load().miso_pair_latent(C.byref(g), ops._ptr(pose.contiguous()), ops._ptr(coords_src),
                                            ops._ptr(feats_src), feats_src.stride(0) if n else n_ch, n, int(loss_type),
                                            ops._ptr(out), ops._stream(coords_src)), "miso_pair_latent")
    return out


@pair_latent_fwd_bwd.register_fake
def _(pose, coords_src, feats_src, feats_dst, bound, ignore_mask, flags, loss_type):
    return torch.empty(24, device=coords_src.device, dtype=torch.float64)


@torch.library.custom_op(f"{_LIB}::lm_normal_eq", mutates_args=(), device_types="cuda")
def lm_normal_eq(coords_frame: torch.Tensor, R_frame: torch.Tensor, grad_world: torch.Tensor, sdf_pred: torch.Tensor,
                 sdf_gt: torch.Tensor, loss_type: int, gm_scale: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> H (6,6), g (6,1), sum w r^2 (Tracker.lm_step); loss_type 2 = L2, 3 = GM"""
    H, g, s = ops.lm_normal_eq(coords_frame, R_frame, grad_world, sdf_pred, sdf_gt, {2: "L2", 3: "GM"}[int(loss_type)],
                               gm_scale)
    return H, g.clone(), s.clone()


@lm_normal_eq.register_fake
def _(coords_frame, R_frame, grad_world, sdf_pred, sdf_gt, loss_type, gm_scale):
    e = coords_frame.new_empty
    return e((6, 6)), e((6, 1)), e(())


@torch.library.custom_op(f"{_LIB}::adam_dense", mutates_args=("param", "grad", "exp_avg", "exp_avg_sq"),
                         device_types="cuda")
def adam_dense(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
               lr: float, beta1: float, beta2: float, eps: float, zero_grad: bool) -> None:
    """torch.optim.Adam (amsgrad=False, weight_decay=0) in place; zero_grad also clears ``grad``."""
    ops.adam_dense_(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, zero_grad)


OPS = ("encode_fwd", "encode_bwd", "encode_bwd2", "encode_decode_fwd", "decode_bwd", "pair_latent_fwd_bwd",
       "lm_normal_eq", "adam_dense")
