"""Operators of the MISO hot path on MI355X: thin autograd wrappers over the
C ABI of libmiso_hip.so (include/miso_hip.h).

* ``encode``      multi-level trilinear feature lookup, differentiable to second
                  order (mirrors third_party/cuda_gridsample_grad2/cuda_gridsample.py:76-126:
                  a forward Function whose backward is itself a Function).
* ``sdf_fused``   encode + frozen decoder MLP in one kernel (GridNet.forward,
                  grid_opt/models/grid_net.py:306-325), first-order fused backward.
* ``adam_dense_`` dense Adam step (grid_opt/trainer.py:217).

There is NO CPU / PyTorch fallback in this module: tensors must live on a HIP
device and the library must be built, otherwise the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib


@dataclass(frozen=True)
class GridMeta:
    """Host-side description shared by all levels of one submap."""
    bound_min: Tuple[float, float, float]
    bound_max: Tuple[float, float, float]
    ignore_mask: int = 0
    flags: int = 0

    @staticmethod
    def from_bound(bound, ignore_level=None, flags: int = 0) -> "GridMeta":
        """bound: (3,2) tensor / nested list, rows = x,y,z [min,max] (base_net.py:25-30)."""
        b = bound.detach().cpu().tolist() if isinstance(bound, torch.Tensor) else bound
        m = 0
        if ignore_level is not None:
            for l, ig in enumerate(ignore_level):
                if bool(ig):
                    m |= 1 << l
        return GridMeta((float(b[0][0]), float(b[1][0]), float(b[2][0])),
                        (float(b[0][1]), float(b[1][1]), float(b[2][1])), m, flags)


NORMALIZED = GridMeta((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), 0, _lib.F_COORDS_NORMALIZED)

# Decoder arithmetic of the fused encode + decoder entries (include/miso_hip.h: MISO_F_EXACT_F32).  Default: bf16x3 split
# products with fp32 accumulation; exact_fp32(True) makes every later call of this process use the exact fp32 chains of
# rounds 1-5 (a launch captured in a HIP graph keeps the form it was captured with).  A single submap can ask for the exact
# form through GridMeta.flags | F_EXACT_F32 as well.
_EXACT_F32 = False


def set_exact_fp32(on: bool) -> bool:
    """Switch the decoder arithmetic for all later fused calls; returns the previous setting."""
    global _EXACT_F32
    prev, _EXACT_F32 = _EXACT_F32, bool(on)
    return prev


class exact_fp32:
    """with ops.exact_fp32(): ... -- the exact fp32 decoder chains inside the block."""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = set_exact_fp32(self.on)
        return self

    def __exit__(self, *exc):
        set_exact_fp32(self.prev)
        return False


def _require_hip(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("miso_amd ops run on the HIP device only (no CPU fallback); "
                               f"got a tensor on {t.device}")
        if t.dtype != torch.float32:
            raise RuntimeError(f"miso_amd ops are fp32 (the reference hard-wires float32); got {t.dtype}")


def _stream(t: torch.Tensor):
    # (torch.cuda.current_stream(...).cuda_stream builds a Stream object: 7 us a call; this is the raw handle)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(t.device.index if t.device.index is not None
                                                         else torch.cuda.current_device()))


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _fill_grid(features: Sequence[torch.Tensor], meta: GridMeta,
               grads: Optional[Sequence[Optional[torch.Tensor]]] = None,
               data: bool = True, touched: Optional[Sequence[Optional[torch.Tensor]]] = None) -> _lib.Grid:
    """touched: per level, the uint8 flags of adam_active_flags() that the scatter kernels set where they put a
    non-zero into ``grads[l]`` (miso_level_t.grad_touched) -- the gradient must then be a dense storage."""
    if not 1 <= len(features) <= _lib.MAX_LEVELS:
        raise ValueError(f"1..{_lib.MAX_LEVELS} levels supported, got {len(features)}")
    g = _lib.Grid()
    g.n_levels = len(features)
    g.ignore_mask = meta.ignore_mask
    g.flags = meta.flags | (_lib.F_EXACT_F32 if _EXACT_F32 else 0)
    for a in range(3):
        g.bound_min[a] = meta.bound_min[a]
        g.bound_max[a] = meta.bound_max[a]
    for l, f in enumerate(features):
        assert f.ndim == 5 and f.shape[0] == 1, f"feature must be (1,C,Z,Y,X), got {tuple(f.shape)}"
        lv = g.level[l]
        lv.data = f.data_ptr() if data else 0
        gr = None if grads is None else grads[l]
        if gr is not None:
            assert gr.shape == f.shape and gr.stride() == f.stride(), "grad must share the feature layout"
            lv.grad = gr.data_ptr()
            tc = None if touched is None else touched[l]
            if tc is not None:
                assert tc.dtype == torch.uint8 and tc.numel() * _lib.ADAM_CHUNK >= gr.numel() and tc.is_contiguous()
                assert gr.is_contiguous() or gr.is_contiguous(memory_format=torch.channels_last_3d), \
                    "touched flags index the gradient's dense storage"
                lv.grad_touched = tc.data_ptr()
            else:
                lv.grad_touched = 0
        else:
            lv.grad = 0
            lv.grad_touched = 0
        lv.C, lv.Z, lv.Y, lv.X = f.shape[1], f.shape[2], f.shape[3], f.shape[4]
        lv.sC, lv.sZ, lv.sY, lv.sX = f.stride(1), f.stride(2), f.stride(3), f.stride(4)
    return g


def _rows(t: torch.Tensor) -> torch.Tensor:
    """(N,K) fp32 with unit inner stride (row stride passed to the kernel)."""
    if t.stride(-1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _feature_dim(features) -> int:
    return sum(int(f.shape[1]) for f in features)


# --------------------------------------------------------------------------- #
# raw calls
# --------------------------------------------------------------------------- #
def encode_fwd_raw(x, features, meta: GridMeta, sorted_batch: Optional["SortedBatch"] = None) -> torch.Tensor:
    """sorted_batch: a SortedBatch already sorted for these points (gathers then run in tile order)."""
    _require_hip(x, *features)
    x = x.contiguous()
    n = x.shape[0]
    out = torch.empty((n, _feature_dim(features)), device=x.device, dtype=torch.float32)
    g = _fill_grid(features, meta)
    ld = out.stride(0) if n else out.shape[1]
    if sorted_batch is not None:
        _lib.check(_lib.load().miso_encode_fwd_sorted(C.byref(g), C.byref(sorted_batch.struct), n, _ptr(out), ld,
                                                      _stream(x)), "miso_encode_fwd_sorted")
    else:
        _lib.check(_lib.load().miso_encode_fwd(C.byref(g), _ptr(x), n, _ptr(out), ld, _stream(x)),
                   "miso_encode_fwd")
    return out


# Batch size from which the grid half of the encode backward bins the batch and pulls
# (sort ~25 us + pull ~110 us at 262144 points, against ~9.3 ns per point of float atomics:
# 2.4 ms at 262144).  None = always atomics.
ENCODE_PULL_MIN_POINTS = 16384


def encode_pull_applies(n: int, meta: GridMeta) -> bool:
    """Whether a batch of n points takes the binned path (sort once in the forward, tile-ordered
    gathers, owner-computes grid gradient in the backward)."""
    return (ENCODE_PULL_MIN_POINTS is not None and n >= ENCODE_PULL_MIN_POINTS
            and not meta.flags & (_lib.F_ALIGN_CORNERS | _lib.F_PAD_BORDER))


def encode_bwd_raw(x, features, meta: GridMeta, gout, need_x: bool, need_f: Sequence[bool],
                   sorted_batch: Optional["SortedBatch"] = None):
    """sorted_batch: the SortedBatch of the forward, if it binned the batch (saves the sort)."""
    _require_hip(x, gout, *features)
    x = x.contiguous()
    gout = _rows(gout)
    n = x.shape[0]
    lib = _lib.load()
    pulled = 0
    grads: List[Optional[torch.Tensor]] = [None] * len(features)
    if (ENCODE_PULL_MIN_POINTS is not None and n >= ENCODE_PULL_MIN_POINTS and any(need_f)
            and gout.stride(0) % 4 == 0 and gout.data_ptr() % 16 == 0):
        want = [torch.empty_like(f) if nf else None for f, nf in zip(features, need_f)]
        pulled = int(lib.miso_grad_pull_levels(C.byref(_fill_grid(features, meta, want, data=False)),
                                               SortedBatch.TILES))
        if pulled:
            if sorted_batch is None:
                sorted_batch = SortedBatch(n, x.device).sort(x, meta)
            mine = [w if (pulled >> l) & 1 else None for l, w in enumerate(want)]
            grad_pull_raw(features, meta, sorted_batch, gout, mine, overwrite=True, caller_order=True)
            grads = mine
    rest = [bool(nf) and not (pulled >> l) & 1 for l, nf in enumerate(need_f)]
    gx = None
    if need_x or any(rest):
        for l, r in enumerate(rest):
            if r:
                grads[l] = torch.zeros_like(features[l])
        gx = torch.empty((n, 3), device=x.device, dtype=torch.float32) if need_x else None
        g = _fill_grid(features, meta, [gr if r else None for gr, r in zip(grads, rest)])
        ld = gout.stride(0) if n else _feature_dim(features)
        if sorted_batch is not None:      # the batch is binned already: gather in tile order
            _lib.check(lib.miso_encode_bwd_sorted(C.byref(g), C.byref(sorted_batch.struct), n, _ptr(gout), ld, _ptr(gx),
                                                  _stream(x)), "miso_encode_bwd_sorted")
        else:
            _lib.check(lib.miso_encode_bwd(C.byref(g), _ptr(x), n, _ptr(gout), ld, _ptr(gx), _stream(x)),
                       "miso_encode_bwd")
    return gx, grads


def encode_bwd2_raw(x, features, meta: GridMeta, gout, ggx, ggf, need_x: bool, need_f: Sequence[bool],
                    sorted_batch: Optional["SortedBatch"] = None):
    """Second backward (gridsample_grad2.grad2_3d).  Large batches: the grid-gradient output is
    formed by the owner-computes pull (miso_grad_pull_dx) instead of float atomics -- they were
    2.4 of the 2.5 ms of this op at 262144 points."""
    _require_hip(x, gout, ggx, *features)
    x = x.contiguous()
    gout = _rows(gout)
    n = x.shape[0]
    F = _feature_dim(features)
    want = [bool(nf) and ggx is not None for nf in need_f]
    grads: List[Optional[torch.Tensor]] = [None] * len(features)
    pulled = 0
    if (any(want) and encode_pull_applies(n, meta) and gout.stride(0) % 4 == 0 and gout.data_ptr() % 16 == 0):
        cand = [torch.empty_like(f) if w else None for f, w in zip(features, want)]
        pulled = int(_lib.load().miso_grad_pull_levels(C.byref(_fill_grid(features, meta, cand, data=False)),
                                                       SortedBatch.TILES))
        if pulled:
            sorted_batch = sorted_batch if sorted_batch is not None else SortedBatch(n, x.device).sort(x, meta)
            mine = [c if (pulled >> l) & 1 else None for l, c in enumerate(cand)]
            gp = _fill_grid(features, meta, mine, data=False)
            gp.flags |= _lib.F_GRAD_OVERWRITE
            _lib.check(_lib.load().miso_grad_pull_dx(C.byref(gp), C.byref(sorted_batch.struct), n, _ptr(gout), gout.stride(0),
                                                     _ptr(ggx.contiguous()), _stream(x)), "miso_grad_pull_dx")
            grads = mine
    for l, w in enumerate(want):
        if w and not (pulled >> l) & 1:
            grads[l] = torch.zeros_like(features[l])
    g = _fill_grid(features, meta, [gr if not (pulled >> l) & 1 else None for l, gr in enumerate(grads)])
    gg = None
    keep = []
    if ggf is not None and any(t is not None for t in ggf):
        gg = _fill_grid(features, meta, None, data=False)
        for l, t in enumerate(ggf):
            if t is None:
                continue
            _require_hip(t)
            if t.stride() != features[l].stride():
                t = torch.empty_like(features[l]).copy_(t)
            keep.append(t)
            gg.level[l].data = t.data_ptr()
    if ggx is not None:
        ggx = ggx.contiguous()
    gg_out = torch.empty((n, F), device=x.device, dtype=torch.float32)
    g_x = torch.empty((n, 3), device=x.device, dtype=torch.float32) if need_x else None
    sb = sorted_batch      # a binned copy exists (the forward's, or the pull's above): gather in tile order as well
    if sb is not None:
        _lib.check(_lib.load().miso_encode_bwd2_sorted(
            C.byref(g), C.byref(gg) if gg is not None else None, C.byref(sb.struct), n, _ptr(gout),
            gout.stride(0) if n else F, _ptr(ggx), _ptr(gg_out), F, _ptr(g_x), _stream(x)), "miso_encode_bwd2_sorted")
    else:
        _lib.check(_lib.load().miso_encode_bwd2(
            C.byref(g), C.byref(gg) if gg is not None else None, _ptr(x), n, _ptr(gout),
            gout.stride(0) if n else F, _ptr(ggx), _ptr(gg_out), F, _ptr(g_x), _stream(x)), "miso_encode_bwd2")
    return gg_out, g_x, grads


# --------------------------------------------------------------------------- #
# encode: autograd to second order
# --------------------------------------------------------------------------- #
class _EncodeBackward(torch.autograd.Function):
    """First backward as a Function so that create_graph=True works (the role of
    _GridSample3dBackward, cuda_gridsample.py:99-126)."""

    @staticmethod
    def forward(ctx, gout, x, meta, need_x, need_f, *features):
        sb = None
        if isinstance(meta, tuple):      # (GridMeta, SortedBatch of the forward)
            meta, sb = meta
        gx, grads = encode_bwd_raw(x, features, meta, gout, need_x, need_f, sorted_batch=sb)
        ctx.save_for_backward(gout, x, *features)
        ctx.meta = meta
        ctx.sorted = sb
        # an eikonal loss differentiates grad_x only: without this autograd hands the second backward a zero tensor the
        # size of every level as the cotangent of its (unused) grid gradient, filled and then gathered eight corners a point
        ctx.set_materialize_grads(False)
        return (gx, *grads)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ggx, *ggf):
        gout, x, *features = ctx.saved_tensors
        need_gout, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_f = ctx.needs_input_grad[5:]
        if ggx is None and all(t is None for t in ggf):
            return (None,) * (5 + len(features))
        gg_out, g_x, g_f = encode_bwd2_raw(x, features, ctx.meta, gout, ggx, ggf, need_x, need_f,
                                           sorted_batch=ctx.sorted)
        return (gg_out if need_gout else None, g_x, None, None, None, *g_f)


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, meta, *features):
        sb = None
        if encode_pull_applies(x.shape[0], meta) and any(ctx.needs_input_grad[2:]):
            # the backward will bin the batch for the pull anyway: bin now and gather in tile order
            sb = SortedBatch(x.shape[0], x.device).sort(x, meta)
        out = encode_fwd_raw(x, features, meta, sorted_batch=sb)
        ctx.save_for_backward(x, *features)
        ctx.meta = meta
        ctx.sorted = sb
        return out

    @staticmethod
    def backward(ctx, gout):
        x, *features = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        need_f = tuple(ctx.needs_input_grad[2:])
        if _ONLY_X and torch.is_grad_enabled():
            need_f = (False,) * len(features)          # (coordinate_gradient_only: see there)
        meta = ctx.meta if ctx.sorted is None else (ctx.meta, ctx.sorted)
        res = _EncodeBackward.apply(gout, x, meta, need_x, need_f, *features)
        return (res[0], None, *res[1:])


def encode(x: torch.Tensor, features: Sequence[torch.Tensor], meta: GridMeta) -> torch.Tensor:
    """(N,3) metres -> (N, sum C) interpolated features of every level, concatenated
    (utils.grid_interp_regular, grid_opt/utils/utils.py:143-164)."""
    return _Encode.apply(x, meta, *features)


def grid_sample_3d(input: torch.Tensor, grid: torch.Tensor, padding_mode: str = "zeros",
                   align_corners: bool = True) -> torch.Tensor:
    """Drop-in for cuda_gridsample.grid_sample_3d (cuda_gridsample.py:17-19) for the
    batch-1 call shape MISO uses: input (1,C,D,H,W), grid (1,Do,Ho,Wo,3) -> (1,C,Do,Ho,Wo)."""
    assert padding_mode in ("zeros", "border")
    assert input.ndim == 5 and grid.ndim == 5 and grid.shape[4] == 3
    assert input.shape[0] == grid.shape[0]
    if input.shape[0] != 1:
        raise RuntimeError("miso_amd.grid_sample_3d supports batch size 1 (MISO's call shape)")
    flags = _lib.F_COORDS_NORMALIZED
    if align_corners:
        flags |= _lib.F_ALIGN_CORNERS
    if padding_mode == "border":
        flags |= _lib.F_PAD_BORDER
    meta = GridMeta((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), 0, flags)
    _, do, ho, wo, _ = grid.shape
    out = _Encode.apply(grid.reshape(-1, 3), meta, input)          # (N,C)
    return out.transpose(0, 1).reshape(1, input.shape[1], do, ho, wo)


# --------------------------------------------------------------------------- #
# fused encode + frozen decoder
# --------------------------------------------------------------------------- #
class DecoderPack:
    """Weights of a frozen MLPNet pre-permuted into MFMA operand order
    (miso_mlp_pack).  Re-packed automatically when a weight tensor changes."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]]):
        self.weights = list(weights)
        self.biases = list(biases)
        self._key = None
        self._packed = None
        self._mlp = None
        self._keep = None

    def _struct(self):
        ws = [w.detach().contiguous() for w in self.weights]
        bs = [None if b is None else b.detach().contiguous() for b in self.biases]
        m = _lib.Mlp()
        m.n_linear = len(ws)
        if not 2 <= m.n_linear <= _lib.MAX_LINEAR:
            return None, None
        m.in_dim = ws[0].shape[1]
        m.hidden_dim = ws[0].shape[0]
        m.out_dim = ws[-1].shape[0]
        for i in range(1, len(ws) - 1):
            if tuple(ws[i].shape) != (m.hidden_dim, m.hidden_dim):
                return None, None
        if ws[-1].shape[1] != m.hidden_dim:
            return None, None
        for i, (w, b) in enumerate(zip(ws, bs)):
            m.weight[i] = w.data_ptr()
            m.bias[i] = 0 if b is None else b.data_ptr()
        return m, (ws, bs)

    def get(self):
        """-> (Mlp struct, packed tensor) or (None, None) if the shape is not covered."""
        key = tuple((w.data_ptr(), w._version) for w in self.weights) + \
            tuple((0, 0) if b is None else (b.data_ptr(), b._version) for b in self.biases)
        if key == self._key:
            return self._mlp, self._packed
        _require_hip(*self.weights)
        m, keep = self._struct()
        self._key, self._mlp, self._packed, self._keep = key, None, None, None
        if m is None:
            return None, None
        nf = _lib.load().miso_mlp_packed_floats(C.byref(m))
        if nf == 0:
            return None, None
        packed = torch.empty(nf, device=self.weights[0].device, dtype=torch.float32)
        _lib.check(_lib.load().miso_mlp_pack(C.byref(m), _ptr(packed), _stream(packed)), "miso_mlp_pack")
        self._mlp, self._packed, self._keep = m, packed, keep
        return m, packed


def sdf_fused_supported(features, meta: GridMeta, pack: DecoderPack) -> bool:
    if not all(f.is_cuda for f in features):
        return False
    m, packed = pack.get()
    if m is None:
        return False
    g = _fill_grid(features, meta)
    return bool(_lib.load().miso_sdf_supported(C.byref(g), C.byref(m)))


def pack_tiles(tiles) -> int:
    """include/miso_hip.h MISO_TILES_XYZ: a count stays a count, (tx, ty, tz) is packed (a cubic one of <= 16 as a count)."""
    if isinstance(tiles, (tuple, list)):
        tx, ty, tz = (int(v) for v in tiles)
        assert all(1 <= v <= 32 for v in (tx, ty, tz)), tiles
        if tx == ty == tz and tx <= 16:
            return tx
        return tx | (ty << 8) | (tz << 16)
    return int(tiles)


def n_tiles(code: int) -> int:
    return code ** 3 if code < 256 else (code & 255) * ((code >> 8) & 255) * ((code >> 16) & 255)


def choose_tiles(features) -> int:
    """Binning for a training step over these levels ((1,C,Z,Y,X) tensors): 16 tiles per axis, and more (up to 32) on an
    axis where the finest level would otherwise put more than 8 vertices on a tile -- what the owner-computes gradient can
    own (ScanNet's 200 x 100 x 200 level: (25, 16, 25)).  A level too fine even for 32 tiles stays scattered."""
    t = []
    for a in (4, 3, 2):                       # x, y, z
        sizes = [int(f.shape[a]) for f in features]
        need = -(-max(sizes) // 8)
        # the matrix-core pull (the only kernel of a binning finer than 16) wants 3 size >= 2 tiles of every level
        cap = (3 * min(sizes)) // 2
        t.append(need if 16 < need <= min(32, cap) else 16)
    return pack_tiles(t)


class SortedBatch:
    """A point batch binned by coarse spatial tile (miso_sort_points).  Buffers are
    reused across calls with the same n (graph-capture friendly)."""

    TILES = 16         # tiles per axis
    # Batch size from which the autograd path bins automatically when grid gradients are wanted.
    # Measured on MI355X (cfg-2, 262144 points): binning costs ~43 us (first version), the forward gathers drop
    # from 76 to 47 us, and the backward goes from 240 us (atomic scatter + 15 us zero-fill) to
    # 157 us (MFMA pass + owner-computes pull, no zero-fill).  Below ~64 K points the tiles hold
    # too few points for the sweep to pay.  None = never automatic.
    AUTO_MIN_POINTS = int(os.environ.get("MISO_SORT_MIN_POINTS", 65536)) or None

    def __init__(self, n: int, device, tiles=TILES, keep_metric: bool = False, need_perm: bool = True):
        """tiles: tiles per axis -- a count (1..16) or per-axis counts (tx, ty, tz) of 1..32 (MISO_TILES_XYZ).
        need_perm=False: no perm[] array (the index rides in xn_sorted[:, 3]); only sdf_train_raw takes such a batch."""
        self.n, self.tiles = int(n), pack_tiles(tiles)
        i32 = dict(device=device, dtype=torch.int32)
        # the kernels read the normalised float4 copy; the metric copy is optional
        self.x_sorted = torch.empty((self.n, 3), device=device, dtype=torch.float32) if keep_metric else None
        self.xn_sorted = torch.empty((self.n, 4), device=device, dtype=torch.float32)
        self.perm = torch.empty(self.n, **i32) if need_perm else None
        self.tile_offsets = torch.empty(n_tiles(self.tiles) + 1, **i32)
        ws = _lib.load().miso_sort_workspace_bytes(self.n, self.tiles)
        self.workspace = torch.empty(max(ws, 1), device=device, dtype=torch.uint8)
        self.struct = _lib.Sorted()
        self.struct.tiles_per_axis = self.tiles
        self.struct.x_sorted = self.x_sorted.data_ptr() if keep_metric else None
        self.struct.xn_sorted = self.xn_sorted.data_ptr()
        self.struct.perm = self.perm.data_ptr() if need_perm else None
        self.struct.tile_offsets = self.tile_offsets.data_ptr()
        # slice queue of the owner-computes gradient (heavy tiles are cut and spread): zeroed once, the
        # library rewinds it after every use
        self.pull_queue = torch.zeros(int(_lib.load().miso_pull_queue_ints(self.n)), **i32)
        self.struct.pull_queue = self.pull_queue.data_ptr()
        self.struct.pull_queue_ints = self.pull_queue.numel()

    def bwd_workspace(self, floats: int) -> torch.Tensor:
        """(N,F) d-feat rows handed from the MFMA backward to the per-tile reduction."""
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < floats:
            ws = torch.empty(max(floats, 4), device=self.xn_sorted.device, dtype=torch.float32)
            self._ws = ws
        return ws

    def sort(self, x: torch.Tensor, meta: GridMeta):
        _require_hip(x)
        x = x.contiguous()
        assert x.shape == (self.n, 3)
        g = _lib.Grid()
        g.n_levels = 1
        g.flags = meta.flags
        for a in range(3):
            g.bound_min[a] = meta.bound_min[a]
            g.bound_max[a] = meta.bound_max[a]
        lv = g.level[0]
        lv.C = lv.X = lv.Y = lv.Z = 1
        lv.sC = lv.sX = lv.sY = lv.sZ = 1
        _lib.check(_lib.load().miso_sort_points(C.byref(g), _ptr(x), self.n, self.tiles, _ptr(self.workspace),
                                                _ptr(self.x_sorted), _ptr(self.xn_sorted), _ptr(self.perm),
                                                _ptr(self.tile_offsets), _stream(x)), "miso_sort_points")
        return self


def sdf_fwd_raw(x, features, meta, pack: DecoderPack, want_mask: bool, out=None, mask=None,
                sorted_batch: Optional[SortedBatch] = None):
    """sorted_batch: a SortedBatch already sorted for these points (x is then ignored)."""
    _require_hip(x, *features)
    m, packed = pack.get()
    if m is None:
        raise RuntimeError("decoder shape is not covered by the fused kernels")
    x = x.contiguous()
    n = x.shape[0]
    sdf = torch.empty((n, 1), device=x.device, dtype=torch.float32) if out is None else out
    if want_mask and mask is None:
        mw = _lib.load().miso_sdf_mask_words(C.byref(m))
        mask = torch.empty(((n + 63) // 64) * 64 * mw, device=x.device, dtype=torch.int32)
    if not want_mask:
        mask = None
    g = _fill_grid(features, meta)
    if sorted_batch is not None:
        _lib.check(_lib.load().miso_sdf_fwd_sorted(C.byref(g), C.byref(m), _ptr(packed),
                                                   C.byref(sorted_batch.struct), n, _ptr(sdf), _ptr(mask),
                                                   _stream(x)), "miso_sdf_fwd_sorted")
    else:
        _lib.check(_lib.load().miso_sdf_fwd(C.byref(g), C.byref(m), _ptr(packed), _ptr(x), n, _ptr(sdf),
                                            _ptr(mask), _stream(x)), "miso_sdf_fwd")
    return sdf, mask


def sdf_bwd_raw(x, features, meta, pack: DecoderPack, gsdf, mask, need_x, need_f, grads=None,
                sorted_batch: Optional[SortedBatch] = None, overwrite: bool = False, gsdf_sorted: bool = False,
                touched: Optional[Sequence[Optional[torch.Tensor]]] = None, zeroed: bool = False):
    """overwrite (binned path only): the gradients are written, not accumulated -- ``grads``
    need no zero-fill (MISO_F_GRAD_OVERWRITE).  gsdf_sorted (binned path only): ``gsdf`` is in
    the binned order (sdf_fwd_loss_raw), not the caller's.  touched: see _fill_grid / adam_active_.
    zeroed (with overwrite): the levels the call adds to with atomics (sdf_bwd_scattered_levels) are zero already --
    the Adam launch that consumed them cleared them -- so the library's fill is skipped (MISO_F_GRAD_ZEROED)."""
    _require_hip(x, gsdf, *features)
    m, packed = pack.get()
    x = x.contiguous()
    gsdf = gsdf.contiguous()
    n = x.shape[0]
    overwrite = overwrite and sorted_batch is not None
    if grads is None:
        alloc = torch.empty_like if overwrite else torch.zeros_like
        grads = [alloc(f) if nf else None for f, nf in zip(features, need_f)]
    gx = torch.empty((n, 3), device=x.device, dtype=torch.float32) if need_x else None
    g = _fill_grid(features, meta, grads, touched=touched)
    if overwrite:
        g.flags |= _lib.F_GRAD_OVERWRITE
        if zeroed:
            g.flags |= _lib.F_GRAD_ZEROED
    if gsdf_sorted:
        assert sorted_batch is not None
        g.flags |= _lib.F_GRAD_SDF_SORTED
    if sorted_batch is not None:
        ws = sorted_batch.bwd_workspace(n * _feature_dim(features)) if any(gr is not None for gr in grads) else None
        _lib.check(_lib.load().miso_sdf_bwd_sorted(C.byref(g), C.byref(m), _ptr(packed),
                                                   C.byref(sorted_batch.struct), n, _ptr(gsdf), _ptr(mask),
                                                   _ptr(gx), _ptr(ws), _stream(x)), "miso_sdf_bwd_sorted")
    else:
        _lib.check(_lib.load().miso_sdf_bwd(C.byref(g), C.byref(m), _ptr(packed), _ptr(x), n, _ptr(gsdf),
                                            _ptr(mask), _ptr(gx), _stream(x)), "miso_sdf_bwd")
    return gx, grads


def sdf_bwd_rows_raw(x, features, meta, pack: DecoderPack, gsdf, mask, need_x, need_f, grads=None):
    """sdf_bwd_raw (caller-order points) that also returns the d-feat rows (N,F) of the decoder backward
    (miso_sdf_bwd_rows): -> (gx, grads, rows)."""
    _require_hip(x, gsdf, *features)
    m, packed = pack.get()
    x = x.contiguous()
    gsdf = gsdf.contiguous()
    n = x.shape[0]
    if grads is None:
        grads = [torch.zeros_like(f) if nf else None for f, nf in zip(features, need_f)]
    gx = torch.empty((n, 3), device=x.device, dtype=torch.float32) if need_x else None
    rows = torch.empty((n, _feature_dim(features)), device=x.device, dtype=torch.float32)
    g = _fill_grid(features, meta, grads)
    _lib.check(_lib.load().miso_sdf_bwd_rows(C.byref(g), C.byref(m), _ptr(packed), _ptr(x), n, _ptr(gsdf), _ptr(mask),
                                             _ptr(gx), _ptr(rows), _stream(x)), "miso_sdf_bwd_rows")
    return gx, grads, rows


def sdf_bwd_scattered_levels(features, meta, grads, n: int, tiles=None) -> int:
    """Bit l set: sdf_bwd_raw(sorted_batch=..., overwrite=True) forms level l's gradient by adding with atomics (and
    zero-fills it first) rather than by the pull's plain stores (miso_sdf_bwd_scattered_levels).  tiles: the batch's
    binning (SortedBatch.tiles; default the 16-tile one)."""
    g = _fill_grid(features, meta, grads, data=False)
    return int(_lib.load().miso_sdf_bwd_scattered_levels(C.byref(g), pack_tiles(tiles or SortedBatch.TILES), n))


def sdf_mask_words(pack: DecoderPack) -> int:
    """32-bit words of ReLU sign bits per point slot (miso_sdf_mask_words)."""
    m, _ = pack.get()
    if m is None:
        raise RuntimeError("decoder shape is not covered by the fused kernels")
    return int(_lib.load().miso_sdf_mask_words(C.byref(m)))


def sdf_fwd_loss_raw(features, meta, pack: DecoderPack, sorted_batch: SortedBatch, loss_inputs, mask, gsdf_sorted,
                     loss_slots, loss_type="L1", weight_sdf=1.0, weight_fs=0.0, trunc_dist=0.0, sdf_out=None,
                     n_live=None):
    """Binned forward with the mapping loss folded in (miso_sdf_fwd_sorted_loss).  loss_inputs (N,4):
    {target, valid, sign, weight} per point, caller order.  Writes d loss / d sdf in binned order to
    ``gsdf_sorted`` (feed it to sdf_bwd_raw(..., gsdf_sorted=True)) and the per-workgroup loss sums to
    ``loss_slots`` ((LOSS_SLOTS,2) floats, fully overwritten; the loss is loss_slots.sum(0)).
    sdf_out (N,1), caller order, is optional.  n_live: one int32 on the device = live rows of a padded batch
    (the means divide by it instead of N)."""
    _require_hip(loss_inputs, gsdf_sorted, loss_slots, *features)
    if n_live is not None:
        assert n_live.is_cuda and n_live.dtype == torch.int32 and n_live.numel() == 1
    m, packed = pack.get()
    n = sorted_batch.n
    assert loss_inputs.shape == (n, 4) and loss_inputs.is_contiguous()
    assert gsdf_sorted.is_contiguous() and gsdf_sorted.numel() == n
    assert loss_slots.is_contiguous() and loss_slots.numel() == _lib.LOSS_SLOTS * 2
    g = _fill_grid(features, meta)
    _lib.check(_lib.load().miso_sdf_fwd_sorted_loss(
        C.byref(g), C.byref(m), _ptr(packed), C.byref(sorted_batch.struct), n, _LOSS_TYPES[loss_type],
        float(weight_sdf), float(weight_fs), float(trunc_dist), _ptr(loss_inputs), _ptr(sdf_out), _ptr(mask),
        _ptr(gsdf_sorted), _ptr(loss_slots), _ptr(n_live), _stream(gsdf_sorted)), "miso_sdf_fwd_sorted_loss")


def sdf_train_supported(features, meta, grads, pack: Optional["DecoderPack"] = None) -> bool:
    """True when sdf_train_raw / sdf_train_unsorted_raw cover this gradient request: a level with a gradient exists
    (the fused decoder shape itself is sdf_fused_supported's business).  Levels the pull / push cannot form from the
    d-feat rows are scattered with float atomics from the same launch (sdf_train_scattered_levels).
    pack: the decoder (its hidden width and depth size the kernel's LDS; without it 64 x 1, the reference's decoder)."""
    if not any(gr is not None and not (meta.ignore_mask >> l) & 1 for l, gr in enumerate(grads)):
        return False
    #  the scattering variant keeps cell records beside the weights: make sure the widest form of the launch
    # fits a workgroup's LDS, so that a shape that does not is routed to the two-launch path instead of failing there --
    # asked of the library itself (miso_sdf_train_lds_bytes: computed from the kernel's own PackLayout) when the decoder
    # is at hand
    if pack is not None and features[0].is_cuda:
        m, _ = pack.get()
        if m is None:
            return False
        need = int(_lib.load().miso_sdf_train_lds_bytes(C.byref(_fill_grid(features, meta, grads, data=False)), C.byref(m), 1))
        return 0 < need <= LDS_PER_WORKGROUP
    H = pack.weights[0].shape[0] if pack is not None else 64
    NH = len(pack.weights) - 2 if pack is not None else 1
    return sdf_train_lds_bytes(features[0].shape[1], len(features), H, scat=True, hidden_layers=NH) <= LDS_PER_WORKGROUP


LDS_PER_WORKGROUP = 160 * 1024      # gfx950


def sdf_train_lds_bytes(C: int, L: int, H: int, scat: bool, hidden_layers: int = 1) -> int:
    """Dynamic LDS of sdf_train_kernel<C, L, H, NH, SCAT> (sdf_fused.hip: PackLayout + four wavefronts' d-feat tiles
    [64][FP] and, scattering, their cell records [64][L][8] -- two blocks of them where 160 KB allow)."""
    F, RT, KS0, KS1, NH = C * L, H // 32, (C * L + 1) // 2, H // 2, hidden_layers
    exact = KS0 * 64 * RT + 2 * NH * KS1 * 64 * RT + H + NH * H + H + 4 + KS1 * 64
    # the bf16x3 form (decoder.hpp): its matrices as [k-block][row tile][3 pieces][64 lanes][4 dwords] + biases and output
    # weights -- the larger of the two packs for every covered shape, and the one the default launch stages
    smd = lambda kb, rt: kb * rt * 3 * 64 * 4                                           # noqa: E731
    KB0, KBH = (F + 15) // 16, H // 16
    split = (smd(KB0, RT) + NH * smd(KBH, RT) + smd(KBH, RT if NH >= 1 else 1) + max(NH - 1, 0) * smd(KBH, RT)
             + (smd(KBH, 1) if NH >= 1 else 0) + (H + NH * H + H + 4 + 3) // 4 * 4)
    pack = max((exact + 3) // 4 * 4, split)
    FP = (F + 3) // 4 * 4 + 4
    words = pack + 4 * (64 * FP + (64 * L * 8 if scat else 0))
    if scat and F <= 12 and 4 * (words + 4 * 64 * L * 8) <= LDS_PER_WORKGROUP:
        words += 4 * 64 * L * 8          # a second block of cell records where it fits (the rotated loop, launch_train_t)
        eight = pack + 8 * (64 * FP + 2 * 64 * L * 8)          # ... and the eight-wavefront form of it
        if 4 * eight <= LDS_PER_WORKGROUP:
            words = max(words, eight)
    return 4 * words


def sdf_train_scattered_levels(features, meta, grads, tiles=None) -> int:
    """Bit mask of the levels sdf_train_raw scatters with atomics from its kernel (bricks beyond the pull's reach under
    the binning `tiles`: SortedBatch.tiles, default the 16-tile one)."""
    want = sum(1 << l for l, gr in enumerate(grads) if gr is not None and not (meta.ignore_mask >> l) & 1)
    pulled = int(_lib.load().miso_grad_pull_levels(C.byref(_fill_grid(features, meta, grads, data=False)),
                                                   pack_tiles(tiles or SortedBatch.TILES)))
    return want & ~pulled


def sdf_train_raw(features, meta, pack: DecoderPack, sorted_batch: SortedBatch, loss_inputs, loss_slots, grads,
                  loss_type="L1", weight_sdf=1.0, weight_fs=0.0, trunc_dist=0.0, sdf_out=None, n_live=None,
                  touched=None, zeroed: bool = False):
    """One binned training step of a frozen-decoder submap in one library call (miso_sdf_train_sorted): forward +
    mapping loss + decoder backward as ONE launch -- which also scatters the levels the pull cannot form, with float
    atomics -- then the pull / push of the other levels' gradient (overwrite semantics: ``grads`` need no zero-fill).
    Same results as sdf_fwd_loss_raw + sdf_bwd_raw(gsdf_sorted=True, overwrite=True)."""
    _require_hip(loss_inputs, loss_slots, *features)
    if n_live is not None:
        assert n_live.is_cuda and n_live.dtype == torch.int32 and n_live.numel() == 1
    m, packed = pack.get()
    n = sorted_batch.n
    assert loss_inputs.shape == (n, 4) and loss_inputs.is_contiguous()
    assert loss_slots.is_contiguous() and loss_slots.numel() == _lib.LOSS_SLOTS * 2
    g = _fill_grid(features, meta, grads, touched=touched)
    g.flags |= _lib.F_GRAD_OVERWRITE
    if zeroed:
        g.flags |= _lib.F_GRAD_ZEROED
    ws = sorted_batch.bwd_workspace(n * _feature_dim(features))
    _lib.check(_lib.load().miso_sdf_train_sorted(
        C.byref(g), C.byref(m), _ptr(packed), C.byref(sorted_batch.struct), n, _LOSS_TYPES[loss_type], float(weight_sdf),
        float(weight_fs), float(trunc_dist), _ptr(loss_inputs), _ptr(sdf_out), _ptr(loss_slots), _ptr(n_live), _ptr(ws),
        _stream(loss_inputs)), "miso_sdf_train_sorted")


def sdf_train_unsorted_raw(x, features, meta, pack: DecoderPack, loss_inputs, loss_slots, grads, loss_type="L1",
                           weight_sdf=1.0, weight_fs=0.0, trunc_dist=0.0, sdf_out=None, touched=None):
    """One training step of a frozen-decoder submap on an UNBINNED (small) batch in one launch (miso_sdf_train):
    forward + mapping loss + decoder backward + the atomic scatter of every level's gradient, ADDED to ``grads``.
    Same results as sdf_fwd_loss_unsorted_raw + sdf_bwd_raw up to the order of the float atomics."""
    _require_hip(x, loss_inputs, loss_slots, *features)
    m, packed = pack.get()
    n = x.shape[0]
    assert x.is_contiguous() and loss_inputs.shape == (n, 4) and loss_inputs.is_contiguous()
    assert loss_slots.is_contiguous() and loss_slots.numel() == _lib.LOSS_SLOTS * 2
    g = _fill_grid(features, meta, grads, touched=touched)
    _lib.check(_lib.load().miso_sdf_train(
        C.byref(g), C.byref(m), _ptr(packed), _ptr(x), n, _LOSS_TYPES[loss_type], float(weight_sdf), float(weight_fs),
        float(trunc_dist), _ptr(loss_inputs), _ptr(sdf_out), _ptr(loss_slots), _stream(x)), "miso_sdf_train")


def sdf_fwd_loss_unsorted_raw(x, features, meta, pack: DecoderPack, loss_inputs, mask, gsdf, loss_slots, loss_type="L1",
                              weight_sdf=1.0, weight_fs=0.0, trunc_dist=0.0, sdf_out=None):
    """sdf_fwd_loss_raw for an unbinned (small) batch: everything in the caller's order (miso_sdf_fwd_loss)."""
    _require_hip(x, loss_inputs, gsdf, loss_slots, *features)
    m, packed = pack.get()
    n = x.shape[0]
    assert x.is_contiguous() and loss_inputs.shape == (n, 4) and loss_inputs.is_contiguous()
    assert gsdf.is_contiguous() and gsdf.numel() == n and loss_slots.is_contiguous()
    assert loss_slots.numel() == _lib.LOSS_SLOTS * 2
    g = _fill_grid(features, meta)
    _lib.check(_lib.load().miso_sdf_fwd_loss(
        C.byref(g), C.byref(m), _ptr(packed), _ptr(x), n, _LOSS_TYPES[loss_type], float(weight_sdf), float(weight_fs),
        float(trunc_dist), _ptr(loss_inputs), _ptr(sdf_out), _ptr(mask), _ptr(gsdf), _ptr(loss_slots), _stream(x)),
        "miso_sdf_fwd_loss")


def grad_pull_raw(features, meta, sorted_batch: SortedBatch, dfeat, grads, overwrite: bool = True,
                  caller_order: bool = False):
    """Grid gradients from d-feat rows (N,F), owner-computes (miso_grad_pull).  Rows are in binned
    order, or with caller_order in the order of the points handed to SortedBatch.sort."""
    _require_hip(dfeat, *features)
    g = _fill_grid(features, meta, grads, data=False)
    if overwrite:
        g.flags |= _lib.F_GRAD_OVERWRITE
    n = sorted_batch.n
    assert dfeat.stride(-1) == 1
    _lib.check(_lib.load().miso_grad_pull(C.byref(g), C.byref(sorted_batch.struct), n, _ptr(dfeat),
                                          dfeat.stride(0) if dfeat.ndim == 2 else _feature_dim(features),
                                          1 if caller_order else 0, _stream(dfeat)), "miso_grad_pull")
    return grads


_BWD2_TORCH = os.environ.get("MISO_BWD2_TORCH", "0") not in ("", "0")

# torch.autograd.grad(sdf, x, create_graph=True) asks for the coordinate gradient only, but a custom Function's backward
# cannot see which of its inputs the caller listed (ctx.needs_input_grad is fixed at forward time: every grid that
# requires grad reads True) -- it would form the grid gradients too, a zero fill and an atomic scatter of every level,
# to have autograd drop them.  The mirror's own gradient helpers (grid_opt/diff.py, loss_isdf.py, models/encoder.py)
# wrap their call in coordinate_gradient_only(): inside it a differentiable (create_graph) first backward of encode /
# sdf_fused skips the grids.
_ONLY_X = False


class coordinate_gradient_only:
    def __enter__(self):
        global _ONLY_X
        self.prev, _ONLY_X = _ONLY_X, True
        return self

    def __exit__(self, *exc):
        global _ONLY_X
        _ONLY_X = self.prev
        return False


def _mlp_torch(feats, weights, biases):
    h = feats
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = torch.nn.functional.linear(h, w, b)
        if i + 1 < len(weights):
            h = torch.relu(h)
    return h


class _SdfFusedBackward(torch.autograd.Function):
    """The fused first backward as a Function, so that create_graph=True stays inside the library (the role of
    _GridSample3dBackward, cuda_gridsample.py:99-126, for the fused encode + decoder): forward = one launch of
    sdf_bwd_kernel that keeps its d-feat rows; backward (the double backward) = ONE launch of the second-order encode on those
    rows.  d sdf / d x = J_E(x; G)^T rows and the grid gradients scatter(w(x) rows) depend on the decoder through `rows`
    alone, and rows = U(masks) d sdf is piecewise constant in the features (ReLU'' = 0, as autograd has it): nothing of the
    decoder is differentiated twice.  Rounds 1-5 rebuilt the graph from encode + torch.nn.functional.linear here (two rocBLAS
    GEMMs per layer, forward and backward, (N, 64) activations through HBM)."""

    @staticmethod
    def _rows(x, features, meta, pack, gsdf, mask, need_x, need_f, sb):
        """sdf_bwd_rows_raw; with the forward's binned batch ``sb`` (whose ``mask`` is then in the binned order) the launch
        runs on the points in that order -- x and d sdf gathered by perm, d sdf / d x and the rows put back -- instead
        of on a second forward's caller-order sign bits."""
        if sb is None:
            return sdf_bwd_rows_raw(x, features, meta, pack, gsdf, mask, need_x, need_f)
        perm = sb.perm.long()
        gx_s, grads, rows_s = sdf_bwd_rows_raw(x.index_select(0, perm), features, meta, pack,
                                               gsdf.reshape(-1, 1).index_select(0, perm), mask, need_x, need_f)
        gx = None if gx_s is None else torch.empty_like(gx_s).index_copy_(0, perm, gx_s)
        return gx, grads, torch.empty_like(rows_s).index_copy_(0, perm, rows_s)

    @staticmethod
    def forward(ctx, gsdf, x, mask, meta, pack, need_x, need_f, sb, *features):
        gx, grads, rows = _SdfFusedBackward._rows(x, features, meta, pack, gsdf, mask, need_x, need_f, sb)
        ctx.save_for_backward(gsdf, x, mask, rows, *features)
        ctx.meta, ctx.pack, ctx.sb = meta, pack, sb
        ctx.set_materialize_grads(False)      # (see _EncodeBackward: no zero cotangents the size of a level)
        return (gx, *grads)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ggx, *ggf):
        gsdf, x, mask, rows, *features = ctx.saved_tensors
        need_gsdf, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_f = ctx.needs_input_grad[8:]
        if ggx is None and all(t is None for t in ggf):
            return (None,) * (8 + len(features))
        # (the forward's binned batch serves the second-order encode and its pull as well: no second sort)
        gg_rows, g_x, g_f = encode_bwd2_raw(x, features, ctx.meta, rows, ggx, ggf, need_x, need_f, sorted_batch=ctx.sb)
        g_gsdf = None
        if need_gsdf:
            # rows are linear in d sdf: rows = U d sdf with U the rows of a unit cotangent -- one more launch, only when the
            # cotangent of sdf is itself differentiated
            _, _, unit = _SdfFusedBackward._rows(x, features, ctx.meta, ctx.pack, torch.ones_like(gsdf), mask, False,
                                                 [False] * len(features), ctx.sb)
            g_gsdf = (gg_rows * unit).sum(dim=1, keepdim=True).view_as(gsdf)
        return (g_gsdf, g_x, None, None, None, None, None, None, *g_f)


class _SdfFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, meta, pack, *features):
        need = any(ctx.needs_input_grad)
        sb = None
        if (any(ctx.needs_input_grad[3:]) and SortedBatch.AUTO_MIN_POINTS is not None
                and x.shape[0] >= SortedBatch.AUTO_MIN_POINTS):
            # training-size batch: bin the points once, both passes use the binned order
            sb = SortedBatch(x.shape[0], x.device).sort(x, meta)
        sdf, mask = sdf_fwd_raw(x, features, meta, pack, want_mask=need, sorted_batch=sb)
        ctx.save_for_backward(x, mask, *features)
        ctx.meta, ctx.pack, ctx.sb = meta, pack, sb
        return sdf

    @staticmethod
    def backward(ctx, gsdf):
        x, mask, *features = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        need_f = tuple(ctx.needs_input_grad[3:])
        if torch.is_grad_enabled():
            # create_graph=True (eikonal / smoothness terms, loss_isdf.py:367-377): the first backward as a Function whose
            # own backward is the second-order encode (_SdfFusedBackward).  MISO_BWD2_TORCH=1 (dev A/B): the graph rebuilt
            # from encode + torch.nn.functional.linear, as rounds 1-5 did
            if _BWD2_TORCH:
                with torch.enable_grad():
                    out = _mlp_torch(encode(x, features, ctx.meta), ctx.pack.weights, ctx.pack.biases)
                    wanted = ([x] if need_x else []) + [f for f, nf in zip(features, need_f) if nf]
                    got = list(torch.autograd.grad(out, wanted, gsdf, create_graph=True, allow_unused=True))
                gx = got.pop(0) if need_x else None
                gfs = [got.pop(0) if nf else None for nf in need_f]
                return (gx, None, None, *gfs)
            if _ONLY_X:
                need_f = (False,) * len(features)
            # (a forward that binned the batch left its sign bits in the binned order: the first backward then runs in
            # that order too, _SdfFusedBackward._rows)
            res = _SdfFusedBackward.apply(gsdf, x, mask, ctx.meta, ctx.pack, need_x, need_f, ctx.sb, *features)
            return (res[0], None, None, *res[1:])
        gx, grads = sdf_bwd_raw(x, features, ctx.meta, ctx.pack, gsdf, mask, need_x, need_f,
                                sorted_batch=ctx.sb, overwrite=True)
        return (gx, None, None, *grads)


def sdf_fused(x, features, meta: GridMeta, pack: DecoderPack) -> torch.Tensor:
    """(N,3) -> (N,1) SDF with a frozen decoder: one kernel forward, one backward."""
    return _SdfFused.apply(x, meta, pack, *features)


def grid_pool_avg(coords, features, bound_min, cell_size: float, dims):
    """utils.grid_pool_3d_avg on the device in three launches (miso_grid_pool_avg): -> (nx, ny, nz, d)."""
    _require_hip(coords, features)
    coords = coords.contiguous()
    features = features if features.stride(-1) == 1 else features.contiguous()
    n, d = features.shape
    nx, ny, nz = (int(v) for v in dims)
    out = torch.empty((nx, ny, nz, d), device=features.device, dtype=torch.float32)
    cnt = torch.empty(nx * ny * nz, device=features.device, dtype=torch.int32)
    bm = (C.c_float * 3)(*[float(v) for v in bound_min])
    _lib.check(_lib.load().miso_grid_pool_avg(_ptr(coords), _ptr(features), n, d, features.stride(0) if n else d, bm,
                                              float(cell_size), nx, ny, nz, _ptr(out), _ptr(cnt), _stream(features)),
               "miso_grid_pool_avg")
    return out


# --------------------------------------------------------------------------- #
# fused atlas query (GridAtlas.query_feature / forward in one launch)
# --------------------------------------------------------------------------- #
class AtlasQuery:
    """The per-submap loop of GridAtlas.query_feature / forward (grid_opt/models/grid_atlas.py:374-399) as ONE launch
    (miso_atlas_sdf_fwd, csrc/atlas.hip): frame change, bound test, multi-level encode of the points that are inside,
    mean over the submaps that contain a point, submap 0's decoder -- inference only (no autograd).

    features: per submap the list of level tensors (1,C,Z,Y,X), channels-last; metas: per submap its GridMeta (bound).
    The device-resident plan is rebuilt when a feature tensor's storage or a bound changes."""

    def __init__(self):
        self._key = None
        self._plan = None
        self._shape = None

    def _prepare(self, features, metas):
        key = tuple(tuple((f.data_ptr(), tuple(f.shape), f.stride()) for f in fs) for fs in features) + \
            tuple((m.bound_min, m.bound_max, m.flags, m.ignore_mask) for m in metas)
        if key == self._key:
            return
        S = len(features)
        grids = (_lib.Grid * S)()
        for s, (fs, m) in enumerate(zip(features, metas)):
            _require_hip(*fs)
            g = _fill_grid(fs, m)
            C.memmove(C.addressof(grids[s]), C.addressof(g), C.sizeof(_lib.Grid))
        nbytes = int(_lib.load().miso_atlas_plan_bytes(S))
        host = (C.c_char * nbytes)()
        _lib.check(_lib.load().miso_atlas_plan_build(grids, S, C.cast(host, C.c_void_p)), "miso_atlas_plan_build")
        dev = features[0][0].device
        self._plan = torch.frombuffer(host, dtype=torch.uint8).clone().to(dev)
        self._shape = _fill_grid(features[0], metas[0])
        self._key = key

    def __call__(self, features, metas, poses, pack: Optional["DecoderPack"], x=None, axes=None, want_sdf=True,
                 want_feats=False, no_bound=False):
        """poses: (S,12) device floats, per submap R_submap_world row-major then t_submap_world.  x: (N,3) world points,
        or axes = (xs, ys, zs) device vectors of a lattice (point (i,j,k) -> index (i ny + j) nz + k).
        -> (sdf (N,1) or None, feats (N,F) or None)."""
        self._prepare(features, metas)
        S = len(features)
        assert poses.shape == (S, 12) and poses.is_contiguous() and poses.dtype == torch.float32
        _require_hip(poses)
        dev = poses.device
        if x is not None:
            _require_hip(x)
            x = x.contiguous()
            n, dims, ax = x.shape[0], (0, 0, 0), (None, None, None)
        else:
            ax = tuple(a.to(device=dev, dtype=torch.float32).contiguous() for a in axes)
            dims = tuple(int(a.numel()) for a in ax)
            n = dims[0] * dims[1] * dims[2]
        F_ = _feature_dim(features[0])
        sdf = torch.empty((n, 1), device=dev, dtype=torch.float32) if want_sdf else None
        feats = torch.empty((n, F_), device=dev, dtype=torch.float32) if want_feats else None
        m = packed = None
        if want_sdf:
            m, packed = pack.get()
            if m is None:
                raise RuntimeError("decoder shape is not covered by the fused kernels")
        flags = (_lib.F_EXACT_F32 if _EXACT_F32 else 0) | (_lib.F_ATLAS_NO_BOUND if no_bound else 0)
        _lib.check(_lib.load().miso_atlas_sdf_fwd(
            _ptr(self._plan), S, C.byref(self._shape), _ptr(poses), C.byref(m) if m is not None else None, _ptr(packed),
            _ptr(x), n, _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), dims[0], dims[1], dims[2], _ptr(sdf), _ptr(feats), F_,
            flags, _stream(poses)), "miso_atlas_sdf_fwd")
        return sdf, feats


# --------------------------------------------------------------------------- #
# dense Adam
# --------------------------------------------------------------------------- #
def adam_dense_(param, grad, exp_avg, exp_avg_sq, step: int, lr: float, beta1: float = 0.9,
                beta2: float = 0.999, eps: float = 1e-8, zero_grad: bool = False):
    """In-place torch.optim.Adam step (amsgrad=False, weight_decay=0) on one dense tensor."""
    _require_hip(param, grad, exp_avg, exp_avg_sq)
    for t in (grad, exp_avg, exp_avg_sq):
        assert t.shape == param.shape and t.stride() == param.stride(), "Adam state must share the param layout"
    _lib.check(_lib.load().miso_adam_dense(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                           param.numel(), lr, beta1, beta2, eps, step, int(zero_grad),
                                           _stream(param)), "miso_adam_dense")


def adam_active_flags(param) -> torch.Tensor:
    """One byte per ADAM_CHUNK floats of the parameter's storage, zero = never stepped."""
    return torch.zeros((param.numel() + _lib.ADAM_CHUNK - 1) // _lib.ADAM_CHUNK, device=param.device, dtype=torch.uint8)


def adam_active_(param, grad, exp_avg, exp_avg_sq, active, step: int, lr: float, beta1: float = 0.9,
                 beta2: float = 0.999, eps: float = 1e-8, zero_grad: bool = False, guard: Optional[torch.Tensor] = None,
                 touched: Optional[torch.Tensor] = None):
    """adam_dense_ that skips the chunks that cannot move (miso_adam_active): bit-identical results, 4 B per
    element + 28 B per element of the chunks a gradient has ever reached.  guard: device scalar (the step's loss);
    if it is NaN the launch leaves parameters, moments and flags alone (the reference's NaN guard on the device).
    touched: the flags the scatter kernels left for this gradient (sdf_bwd_raw(touched=...)): the launch then reads
    them instead of the gradient (miso_adam_touched) and clears them -- only valid when nothing but those kernels
    wrote the gradient since the last step."""
    _require_hip(param, grad, exp_avg, exp_avg_sq, guard)
    for t in (grad, exp_avg, exp_avg_sq):
        assert t.shape == param.shape and t.stride() == param.stride(), "Adam state must share the param layout"
    assert active.dtype == torch.uint8 and active.numel() * _lib.ADAM_CHUNK >= param.numel()
    if touched is not None:
        assert touched.dtype == torch.uint8 and touched.numel() == active.numel() and touched.is_contiguous()
        _lib.check(_lib.load().miso_adam_touched(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(active),
                                                 _ptr(touched), param.numel(), lr, beta1, beta2, eps, step,
                                                 int(zero_grad), _ptr(guard), _stream(param)), "miso_adam_touched")
        return
    _lib.check(_lib.load().miso_adam_active(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(active),
                                            param.numel(), lr, beta1, beta2, eps, step, int(zero_grad),
                                            _ptr(guard), _stream(param)), "miso_adam_active")


def adam_tensors(tensors):
    """The argument block of adam_active_multi_ / AdamDeviceStep.step_multi_ for ``tensors`` = [(param, grad, exp_avg,
    exp_avg_sq, active, zero_grad[, touched])]: built once for buffers whose addresses do not change.  touched: the flags
    the scatter kernels left for that gradient (adam_active_'s ``touched``), or None.  Returns (block, keep-alive)."""
    assert 1 <= len(tensors) <= _lib.ADAM_MAX_TENSORS
    arr = (_lib.AdamTensor * len(tensors))()
    for a, t7 in zip(arr, tensors):
        p, g, m, v, act, zero = t7[:6]
        tch = t7[6] if len(t7) > 6 else None
        _require_hip(p, g, m, v)
        for t in (g, m, v):
            assert t.shape == p.shape and t.stride() == p.stride(), "Adam state must share the param layout"
        assert act.dtype == torch.uint8 and act.numel() * _lib.ADAM_CHUNK >= p.numel()
        if tch is not None:
            assert tch.dtype == torch.uint8 and tch.numel() == act.numel() and tch.is_contiguous()
        a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
        a.active, a.numel, a.zero_grad = act.data_ptr(), p.numel(), int(bool(zero))
        a.touched = None if tch is None else tch.data_ptr()
    return arr, [tuple(t7[:5]) + ((t7[6],) if len(t7) > 6 else ()) for t7 in tensors]


def adam_active_multi_(packed, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                       guard: Optional[torch.Tensor] = None):
    """adam_active_ for several tensors in ONE launch (miso_adam_active_multi), each bit-identical to its own call;
    ``packed`` from adam_tensors()."""
    arr, keep = packed
    _lib.check(_lib.load().miso_adam_active_multi(arr, len(arr), lr, beta1, beta2, eps, step, _ptr(guard),
                                                  _stream(keep[0][0])), "miso_adam_active_multi")


class HostTotal:
    """A step's loss total as the device hands it to the host (AdamDeviceStep.ring): valid once the slot carries the
    launch's number.  (numpy views of the pinned ring: a read is ~0.1 us, a torch index ~3.)"""
    __slots__ = ("ring", "seqs", "slot", "seq")

    def __init__(self, ring, seqs, slot, seq):
        self.ring, self.seqs, self.slot, self.seq = ring, seqs, slot, seq

    def ready(self) -> bool:
        return int(self.seqs[self.slot, 1]) == self.seq

    def value(self) -> float:
        return float(self.ring[self.slot, 0])


class AdamDeviceStep:
    """What a captured step needs to take Adam steps without new kernel arguments (miso_adam_step_dev): the table of
    per-step scalars (computed by the library's own host code, so the captured step equals the launch-by-launch one
    bit for bit) and the step count on the device.  ``count`` mirrors the device value on the host."""
    MAX_ROWS = 1 << 21

    def __init__(self, lr: float, beta1: float, beta2: float, eps: float, device, count: int = 0):
        import math
        self.hyper = (float(lr), float(beta1), float(beta2), float(eps))
        bmax = max(beta1, beta2)
        # rows until beta^t has left fp32 (the scalars no longer change): 20 700 for beta2 = 0.999
        rows = 64 if bmax <= 0 else int(math.ceil(math.log(1e-9) / math.log(bmax))) + 64 if bmax < 1 else None
        if rows is None or rows > self.MAX_ROWS:
            raise ValueError(f"betas {beta1, beta2}: the step-scalar table would need {rows} rows")
        host = torch.empty((rows, 6), dtype=torch.float32)
        _lib.check(_lib.load().miso_adam_scalars_table(lr, beta1, beta2, eps, 1, rows, C.c_void_p(host.data_ptr())),
                   "miso_adam_scalars_table")
        self.table = host.to(device)
        self.rows = rows
        # [0] the Adam step count, [1] the launches of total_and_bump so far (the slot of ``ring`` a launch writes)
        self.step = torch.zeros(2, dtype=torch.int32, device=device)
        # the step's loss as the host sees it: written by the kernel itself (pinned memory mapped into the device), so
        # the NaN guard costs the stream no copy.  ``launches`` mirrors step[1]: whoever EXECUTES total_and_bump (a
        # stream launch or a graph replay, not a capture) calls note_launch().
        # RING slots of {total, 1-based number of the launch that wrote it}: the host polls the number
        self.ring = torch.zeros((self.RING, 2), dtype=torch.float32).pin_memory()
        self._ring_np = self.ring.numpy()                              # (shares the pinned memory)
        self._seq_np = self.ring.view(torch.int32).numpy()
        self.launches = 0
        self.set_count(count)

    RING = 64

    def set_count(self, count: int):
        self.count = int(count)
        self.step[:1].fill_(self.count)

    def note_launch(self):
        """One total_and_bump has been put on the stream: the HostTotal of the ``ring`` slot it will write (ready() once
        the launch's number stands in word 1, the total then in word 0)."""
        slot = self.launches & (self.RING - 1)
        self.launches += 1
        return HostTotal(self._ring_np, self._seq_np, slot, ((self.launches + 2 ** 31) % 2 ** 32) - 2 ** 31)

    def bump(self, guard: Optional[torch.Tensor]):
        """step += 1 on the device unless ``guard`` (device scalar) is NaN; the host mirror is the caller's business."""
        _lib.check(_lib.load().miso_adam_bump(_ptr(self.step), _ptr(guard), _stream(self.step)), "miso_adam_bump")

    def total_and_bump(self, loss_slots: torch.Tensor, total: torch.Tensor):
        """total (0-d) = loss_slots.sum() and bump(total) in one launch (miso_loss_total_bump)."""
        assert loss_slots.is_contiguous() and loss_slots.dtype == torch.float32 and total.numel() == 1
        _lib.check(_lib.load().miso_loss_total_bump_host(_ptr(loss_slots), loss_slots.numel(), _ptr(total),
                                                         _ptr(self.step), C.c_void_p(self.ring.data_ptr()), self.RING,
                                                         _stream(self.step)), "miso_loss_total_bump_host")

    def step_(self, param, grad, exp_avg, exp_avg_sq, active, touched=None, zero_grad=False, guard=None):
        _require_hip(param, grad, exp_avg, exp_avg_sq, guard)
        _lib.check(_lib.load().miso_adam_step_dev(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(active),
                                                  _ptr(touched), param.numel(), _ptr(self.table), self.rows,
                                                  _ptr(self.step), int(zero_grad), _ptr(guard), _stream(param)),
                   "miso_adam_step_dev")


    def multi(self, tensors):
        """adam_tensors(tensors): the argument block of step_multi_."""
        return adam_tensors(tensors)

    def step_multi_(self, packed, guard=None):
        """step_ for several tensors in one launch (miso_adam_step_dev_multi); ``packed`` from multi()."""
        arr = packed[0]
        _lib.check(_lib.load().miso_adam_step_dev_multi(arr, len(arr), _ptr(self.table), self.rows, _ptr(self.step),
                                                        _ptr(guard), _stream(self.step)), "miso_adam_step_dev_multi")


def mapping_batch(R, t, table, frame_ids, coords_frame, target, valid, sign, weight, x_out, rows_out,
                  sanitize: bool = False):
    """The input side of a mapping step in one launch (miso_mapping_batch): keyframe lookup, frame -> world map and
    the interleaved label rows, written into the step's static buffers ``x_out`` (N,3) / ``rows_out`` (N,4).  The
    label columns may be strided views ((N,1) slices of a row-major block); ``valid`` may be a bool mask.
    sanitize: torch.nan_to_num on every float read (the trainer's prepare_batch folded in).
    Raises ValueError for layouts the launch does not take."""
    n = x_out.shape[0]
    strides = (C.c_int64 * 4)(1, 1, 1, 1)

    # (no views are formed here: a (N,1) column starts where its (N,) slice would, and this function is most of the host's
    # work in a 6 144-sample trainer step -- every tensor view is 1 - 2 us)
    def col(c, i, dtypes=(torch.float32,)):
        if c is None:
            return None
        d = c.dim()
        if not ((d == 1 or (d == 2 and c.shape[1] == 1)) and c.shape[0] == n and c.is_cuda and c.dtype in dtypes
                and (n < 2 or c.stride(0) >= 0)):
            raise ValueError("mapping_batch wants device columns (N,) or (N,1) of the batch's length")
        strides[i] = c.stride(0) if n > 1 else 1
        return c

    cols = [col(target, 0), col(valid, 1, (torch.float32, torch.bool)), col(sign, 2), col(weight, 3)]
    if target is None:
        raise ValueError("mapping_batch needs the target column")
    fid = frame_ids if frame_ids.is_contiguous() else frame_ids.reshape(-1)
    cf = coords_frame if coords_frame.is_contiguous() else coords_frame.reshape(-1, 3)
    if not (fid.dtype == torch.int64 and fid.is_contiguous() and fid.numel() == n and cf.is_contiguous()
            and cf.dtype == torch.float32 and cf.numel() == 3 * n and cf.shape[-1] == 3 and table.dtype == torch.int64
            and fid.is_cuda):
        raise ValueError("mapping_batch: frame ids int64 (N,), coords fp32 (N,3)")
    if not (R.is_contiguous() and t.is_contiguous() and R.dtype == torch.float32 and t.dtype == torch.float32):
        raise ValueError("mapping_batch: poses fp32 contiguous")
    _lib.check(_lib.load().miso_mapping_batch(_ptr(R), _ptr(t), R.shape[0], _ptr(table), table.numel(), _ptr(fid),
                                              _ptr(cf), _ptr(cols[0]), _ptr(cols[1]), _ptr(cols[2]), _ptr(cols[3]),
                                              strides, int(valid is not None and cols[1].dtype == torch.bool), n,
                                              _ptr(x_out), _ptr(rows_out), int(bool(sanitize)), _stream(x_out)),
               "miso_mapping_batch")


# --------------------------------------------------------------------------- #
# mapping loss (value + gradient w.r.t. the prediction)
# --------------------------------------------------------------------------- #
_LOSS_TYPES = {"L1": 1, "L2": 2}


def mapping_loss_raw(pred, target, valid, sign, weight, loss_type: str, weight_sdf: float,
                     weight_fs: float, trunc_dist: float, grad_pred=None, loss_out=None, grad_pred_fs=None):
    """miso_loss_regression + miso_loss_free_space (grid_opt/loss.py:594-635, :668-700) and
    d/d pred in one launch.  Returns (loss_out[2] = weighted terms, grad_pred (N,1))."""
    _require_hip(pred, target, valid, sign, weight)
    n = pred.shape[0]
    pred = pred.contiguous()
    if grad_pred is None:
        grad_pred = torch.empty_like(pred)
    if loss_out is None:
        loss_out = torch.empty(2, device=pred.device, dtype=torch.float32)
    cont = lambda t: None if t is None else t.to(torch.float32).contiguous()
    target, valid, sign, weight = cont(target), cont(valid), cont(sign), cont(weight)
    _lib.check(_lib.load().miso_mapping_loss(_LOSS_TYPES[loss_type], weight_sdf, weight_fs,
                                             0.0 if trunc_dist is None else trunc_dist, _ptr(pred),
                                             _ptr(target), _ptr(valid), _ptr(sign), _ptr(weight), n,
                                             _ptr(grad_pred), _ptr(grad_pred_fs), _ptr(loss_out),
                                             _stream(pred)), "miso_mapping_loss")
    return loss_out, grad_pred


def mapping_loss_rows_raw(pred, rows, loss_type: str, weight_sdf: float, weight_fs: float, trunc_dist: float,
                          grad_pred, loss_out):
    """mapping_loss_raw over the interleaved label rows {target, valid, sign, weight} (N,4) of MappingStep."""
    _require_hip(pred, rows, grad_pred, loss_out)
    assert rows.is_contiguous() and rows.shape == (pred.shape[0], 4) and pred.is_contiguous() and grad_pred.is_contiguous()
    _lib.check(_lib.load().miso_mapping_loss_rows(_LOSS_TYPES[loss_type], weight_sdf, weight_fs,
                                                  0.0 if trunc_dist is None else trunc_dist, _ptr(pred), _ptr(rows),
                                                  pred.shape[0], _ptr(grad_pred), _ptr(loss_out), _stream(pred)),
               "miso_mapping_loss_rows")
    return loss_out, grad_pred


class _MappingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, valid, sign, weight, loss_type, w_sdf, w_fs, trunc):
        gfs = torch.empty_like(pred) if w_fs > 0 else None
        loss, gpred = mapping_loss_raw(pred, target, valid, sign, weight, loss_type, w_sdf, w_fs, trunc,
                                       grad_pred_fs=gfs)
        ctx.save_for_backward(gpred, gfs)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        gpred, gfs = ctx.saved_tensors
        if gfs is None:
            g = gpred * gl[0]
        else:
            g = (gpred - gfs) * gl[0] + gfs * gl[1]
        return (g, None, None, None, None, None, None, None, None)


def mapping_loss(pred, target, valid, sign, weight, loss_type="L1", weight_sdf=1.0, weight_fs=0.0,
                 trunc_dist=0.0):
    """-> tensor([weight_sdf * sdf_term, weight_fs * free_space_term]), differentiable w.r.t. pred.
    The row labels may arrive as bool / integer tensors (the RGB-D dataset's ``sdf_valid`` is bool, sdf_rgbd.py:452)."""
    f32 = lambda t: t if t is None or t.dtype == torch.float32 else t.to(torch.float32)
    valid, sign, weight = f32(valid), f32(sign), f32(weight)
    return _MappingLoss.apply(pred, target, valid, sign, weight, loss_type, weight_sdf, weight_fs,
                              trunc_dist)


# --------------------------------------------------------------------------- #
# latent alignment residual of a submap pair (pose-Jacobian path)
# --------------------------------------------------------------------------- #
class _PairLatent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R_s, t_s, R_d, t_d, coords_src, feats_src, meta_dst, loss_type, *feats_dst):
        _require_hip(R_s, t_s, R_d, t_d, coords_src, feats_src, *feats_dst)
        n = coords_src.shape[0]
        coords_src = coords_src.contiguous()
        feats_src = _rows(feats_src)
        n_ch = _feature_dim(feats_dst)
        assert feats_src.shape[1] >= n_ch
        pose = torch.cat((R_s.reshape(9), t_s.reshape(3), R_d.reshape(9), t_d.reshape(3))).contiguous()
        out = torch.empty(24, device=coords_src.device, dtype=torch.float64)      # the sums arrive in fp64
        g = _fill_grid(feats_dst, meta_dst)
        _lib.check(_lib.load().miso_pair_latent(C.byref(g), _ptr(pose), _ptr(coords_src), _ptr(feats_src),
                                                feats_src.stride(0) if n else n_ch, n, _LOSS_TYPES[loss_type],
                                                _ptr(out), _stream(coords_src)), "miso_pair_latent")
        denom = out[1].clamp(min=1.0) * (n_ch if loss_type == "L2" else 1)
        ctx.save_for_backward(out, denom, R_d)
        return (out[0] / denom).to(R_s.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        out, denom, R_d = ctx.saved_tensors
        s = gl.double() / denom
        h = R_d.double() @ out[2:5]             # sum_i R_dst g_i
        f = lambda t: t.to(R_d.dtype)
        g_Rs = f(out[14:23].view(3, 3) * s)
        g_ts = f(h.view(3, 1) * s)
        g_Rd = f(out[5:14].view(3, 3) * s)
        g_td = f(-h.view(3, 1) * s)
        return (g_Rs, g_ts, g_Rd, g_td) + (None,) * (4 + (len(ctx.needs_input_grad) - 8))


def pair_latent(R_src, t_src, R_dst, t_dst, coords_src, feats_src, feats_dst, meta_dst, loss_type="L2"):
    """mean over in-bound vertices (and channels, L2) of the latent residual between the source
    features and the destination grid sampled at the mapped vertices; differentiable w.r.t. the four
    pose tensors (R (3,3), t (3,1)).  One kernel forward, closed-form backward (miso_pair_latent)."""
    return _PairLatent.apply(R_src, t_src, R_dst, t_dst, coords_src, feats_src, meta_dst, loss_type, *feats_dst)


class _PairLatentMulti(torch.autograd.Function):
    """All pairs of one alignment iteration behind ONE autograd node: P kernel launches, then the
    loss normalisation and the pose cotangents as a handful of batched tensor ops (instead of ~25
    small ops and an autograd node per pair -- the per-pair version is host-bound)."""

    @staticmethod
    def forward(ctx, R_all, t_all, plan):
        lib = _lib.load()
        dev = R_all.device
        S = R_all.shape[0]
        pairs = plan["pairs"]
        P = len(pairs)
        const = plan.get("_const")          # index / count tensors of this pair list: built once, not per iteration
        if const is None or const["dev"] != dev:
            const = dict(dev=dev,
                         src=torch.tensor([a for a, _ in pairs], device=dev, dtype=torch.long),
                         dst=torch.tensor([b for _, b in pairs], device=dev, dtype=torch.long),
                         nch=torch.tensor(plan["n_ch"], device=dev, dtype=torch.float32),
                         npts=(torch.tensor([g_[0].shape[0] for g_ in plan["gate_pts"]], device=dev,
                                            dtype=torch.float32) if plan["gate_pts"] is not None else None))
            plan["_const"] = const
        src, dst = const["src"], const["dst"]
        pose_all = torch.cat((R_all.reshape(S, 9), t_all.reshape(S, 3)), dim=1)            # (S,12)
        pose_pairs = torch.cat((pose_all[src], pose_all[dst]), dim=1).contiguous()           # (P,24)
        out = torch.empty((P, 24), device=dev, dtype=torch.float64)
        cnt = torch.empty(P, device=dev, dtype=torch.float32) if plan["gate_pts"] is not None else None
        stream = _stream(pose_pairs)
        lt = _LOSS_TYPES[plan["loss_type"]]
        for p in range(P):
            coords, fsrc, grid, nch = plan["coords"][p], plan["feats_src"][p], plan["grids"][p], plan["n_ch"][p]
            n = coords.shape[0]
            _lib.check(lib.miso_pair_latent(C.byref(grid), C.c_void_p(pose_pairs.data_ptr() + 96 * p), _ptr(coords),
                                            _ptr(fsrc), fsrc.stride(0) if n else nch, n, lt,
                                            C.c_void_p(out.data_ptr() + 192 * p), stream), "miso_pair_latent")
            if cnt is not None:
                pts, bmin, bmax = plan["gate_pts"][p]
                _lib.check(lib.miso_overlap_count(C.c_void_p(pose_pairs.data_ptr() + 96 * p), _ptr(pts), pts.shape[0],
                                                  bmin, bmax, C.c_void_p(cnt.data_ptr() + 4 * p), stream),
                           "miso_overlap_count")
        denom = out[:, 1].clamp(min=1.0) * (const["nch"].double() if plan["loss_type"] == "L2" else 1.0)
        gate = torch.ones(P, device=dev)
        if cnt is not None:
            gate = ((cnt / const["npts"]) > plan["overlap_thresh"]).to(torch.float32)
        ctx.save_for_backward(out, denom, gate, R_all, src, dst)
        return torch.nan_to_num((out[:, 0] / denom).to(R_all.dtype)) * gate

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        out, denom, gate, R_all, src, dst = ctx.saved_tensors
        # nan_to_num in the forward passes no gradient for a non-finite pair loss (its derivative is
        # grad * isfinite(input)); the raw sums of such a pair may be NaN as well and must not reach the poses
        finite = torch.isfinite((out[:, 0] / denom).to(R_all.dtype))
        out = torch.where(finite.view(-1, 1), out, torch.zeros_like(out))
        s = torch.where(finite, (torch.nan_to_num(gl) * gate).double() / denom, torch.zeros_like(denom)).view(-1, 1, 1)
        h = R_all[dst].double() @ out[:, 2:5].unsqueeze(-1)         # sum_i R_dst g_i, (P,3,1)
        gR = torch.zeros(R_all.shape, device=R_all.device, dtype=torch.float64)
        gt = torch.zeros((R_all.shape[0], 3, 1), device=R_all.device, dtype=torch.float64)
        gR.index_add_(0, src, out[:, 14:23].reshape(-1, 3, 3) * s)
        gR.index_add_(0, dst, out[:, 5:14].reshape(-1, 3, 3) * s)
        gt.index_add_(0, src, h * s)
        gt.index_add_(0, dst, -h * s)
        return gR.to(R_all.dtype), gt.to(R_all.dtype), None


def pair_latent_multi(R_all, t_all, plan) -> torch.Tensor:
    """(P,) pair losses of one alignment iteration (pairwise_loss_latent for every pair of
    ``plan['pairs']``), differentiable w.r.t. the stacked submap poses R_all (S,3,3), t_all (S,3,1).
    plan: dict with per-pair lists ``coords`` (N_p,3), ``feats_src`` (N_p,>=n_ch), ``grids``
    (_lib.Grid of the destination levels), ``n_ch``; ``loss_type``; ``gate_pts`` (per pair
    (finest vertices, c_float[3] bound_min, c_float[3] bound_max)) or None and ``overlap_thresh``:
    pairs whose overlap fraction is not above the threshold contribute exactly 0 -- decided on the
    device (GridAtlas.check_submap_intersection without the host sync)."""
    _require_hip(R_all, t_all)
    return _PairLatentMulti.apply(R_all, t_all, plan)


ALIGN_BOX_VERTS = 64         # MISO_ALIGN_BOX_VERTS


class AlignPlan:
    """Device-resident state of one fused alignment run (miso_align_iteration_a / _b): Adam over the pose
    corrections of submaps 1..S-1 on the summed latent pair losses, generic_align_multiple_submaps
    (grid_opt/align/base.py:89-163) with pairwise_loss_latent (grid_opt/align/miso.py:116-211, L2 / L1).  An
    iteration is ``iteration_a()`` (poses, overlap gates, every pair's residual and pose cotangents, pulled back to
    d loss / d (dr, dt)) then ``iteration_b()`` (regulariser, NaN guard, Adam, early stop, bookkeeping); in between a
    multi-rank caller all-reduces ``flat`` (6S pose gradients + the pair-loss sum).  The host reads nothing inside
    the loop: losses, relative changes and pose snapshots of every iteration sit in ``ring()`` afterwards.

    R0 (S,3,3) / t0 (S,3,1): base poses.  pairs: dicts with ``src``, ``dst``, ``coords`` (n,3), ``feats_src``
    (n, >= F), ``feats_dst`` (list of level tensors, levels 0..level), ``meta_dst`` (GridMeta) and ``gate_pts``
    ((m,3) finest-level vertices of src, or None: no overlap gate) with, optionally, ``gate_dims`` = (nx, ny, nz) when
    those vertices are FeatureGrid.vertex_positions' meshgrid -- the gate then reads three short tables instead of m
    points."""

    def __init__(self, R0, t0, pairs, *, loss_type="L2", align_weight=3000.0, overlap_thresh=1e-2, lr=1e-2,
                 betas=(0.9, 0.999), eps=1e-8, reg_weight=0.0, reg_thresh_rad=1.0, reg_thresh_m=1.0,
                 rel_change_thresh=0.0, ring_iters=0, save_poses=False, cull=None):
        """cull (default on; MISO_ALIGN_CULL=0 switches it off): one box per run of 64 source vertices
        (miso_align_src_boxes, built once per source list) lets the residual kernel skip runs that cannot reach the
        destination bound at the current poses without reading them -- same sums, fewer bytes."""
        _require_hip(R0, t0)
        lib = _lib.load()
        dev = R0.device
        S, P = int(R0.shape[0]), len(pairs)
        self.S, self.P, self.device = S, P, dev
        self._R0 = R0.detach().reshape(S, 9).contiguous().clone()
        self._t0 = t0.detach().reshape(S, 3).contiguous().clone()
        cfg = _lib.Align()
        cfg.n_submaps, cfg.n_pairs, cfg.loss_type = S, P, _LOSS_TYPES[loss_type]
        cfg.ring_iters, cfg.save_poses = int(ring_iters), int(bool(save_poses))
        cfg.align_weight, cfg.overlap_thresh = float(align_weight), float(overlap_thresh)
        cfg.reg_weight, cfg.reg_thresh_rad, cfg.reg_thresh_m = float(reg_weight), float(reg_thresh_rad), float(reg_thresh_m)
        cfg.rel_change_thresh = float(rel_change_thresh)
        cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        self._keep = []
        if cull is None:
            cull = os.environ.get("MISO_ALIGN_CULL", "1") != "0"
        boxes_of = {}                                 # one table per source list (a submap is the source of up to S-1 pairs)
        descs = (_lib.AlignPair * max(P, 1))()
        for i, pr in enumerate(pairs):
            feats_dst = [f.detach() for f in pr["feats_dst"]]
            coords = pr["coords"].detach().contiguous()
            fsrc = _rows(pr["feats_src"].detach())
            _require_hip(coords, fsrc, *feats_dst)
            d = descs[i]
            d.dst_grid = _fill_grid(feats_dst, pr["meta_dst"])
            d.coords_src, d.feats_src = coords.data_ptr(), fsrc.data_ptr()
            d.n = coords.shape[0]
            d.ld_feats = fsrc.stride(0) if d.n else _feature_dim(feats_dst)
            gate = pr.get("gate_pts")
            if gate is not None:
                gate = gate.detach().contiguous()
                _require_hip(gate)
                d.gate_coords, d.gate_n = gate.data_ptr(), gate.shape[0]
                dims = pr.get("gate_dims")           # (nx, ny, nz): the points are a z-major meshgrid of three tables
                if dims is not None:
                    nx, ny, nz = (int(v) for v in dims)
                    assert nx * ny * nz == gate.shape[0]
                    axes = (gate[:nx, 0].contiguous(), gate[::nx][:ny, 1].contiguous(),
                            gate[::nx * ny][:nz, 2].contiguous())
                    # the lattice gate walks rows along x by index arithmetic: the x table must be an increasing,
                    # evenly spaced one (FeatureGrid.vertex_positions: a linspace) -- checked once, here
                    xs = axes[0].double().cpu()
                    even = nx < 2 or bool(((xs[1:] - xs[:-1]) > 0).all() and
                                          ((xs - torch.linspace(float(xs[0]), float(xs[-1]), nx, dtype=torch.float64)).abs()
                                           <= 0.25 * float(xs[-1] - xs[0]) / (nx - 1)).all())
                    if even:
                        for a in range(3):
                            d.gate_axis[a], d.gate_dims[a] = axes[a].data_ptr(), (nx, ny, nz)[a]
                        gate = (gate, axes)
            d.src, d.dst = int(pr["src"]), int(pr["dst"])
            if cull and d.n > 0:
                key = (coords.data_ptr(), int(d.n))
                if key not in boxes_of:
                    bx = torch.empty(((int(d.n) + ALIGN_BOX_VERTS - 1) // ALIGN_BOX_VERTS, 6), device=dev, dtype=torch.float32)
                    _lib.check(lib.miso_align_src_boxes(_ptr(coords), int(d.n), _ptr(bx), _stream(coords)),
                               "miso_align_src_boxes")
                    boxes_of[key] = bx
                d.src_boxes = boxes_of[key].data_ptr()
            self._keep.append((feats_dst, coords, fsrc, gate, boxes_of))
        nbytes = int(lib.miso_align_plan_bytes(P))
        blob = (C.c_uint8 * nbytes)()
        _lib.check(lib.miso_align_plan_build(descs, C.byref(cfg), blob), "miso_align_plan_build")
        self._plan = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        off = (C.c_int64 * 12)()
        total = int(lib.miso_align_state_layout(S, P, cfg.ring_iters, cfg.save_poses, off))
        self._off = [int(v) for v in off]
        self.state = torch.zeros(max(total, 4), device=dev, dtype=torch.float32)
        cfg.R0, cfg.t0 = self._R0.data_ptr(), self._t0.data_ptr()
        cfg.plan, cfg.state = self._plan.data_ptr(), self.state.data_ptr()
        self.cfg = cfg
        self.ring_iters, self.save_poses = cfg.ring_iters, bool(cfg.save_poses)

    def _view(self, k, n):
        return self.state[self._off[k]:self._off[k] + n]

    @property
    def params(self) -> torch.Tensor:
        """(S,6) = (dr, dt) per submap: write the initial corrections here, read the final ones back.  (Handing the
        tensor out counts as a write: the next iteration_a forms the poses itself instead of trusting those the last
        iteration_b left.)"""
        self._poses_ready = False
        return self._view(0, 6 * self.S).view(self.S, 6)

    @property
    def poses(self) -> torch.Tensor:
        """(S,12) = (R row-major, t) as of the last iteration_a."""
        return self._view(1, 12 * self.S).view(self.S, 12)

    @property
    def pair_out(self) -> torch.Tensor:
        """(P,24) fp64 sums of the last iteration_a (miso_pair_latent's `out` per pair)."""
        return self._view(2, 48 * self.P).view(torch.float64).view(self.P, 24)

    @property
    def overlap_counts(self) -> torch.Tensor:
        """(P,) number of gate vertices of src inside dst's bound as of the last iteration_a (float-valued integers)."""
        return self._view(3, self.P)

    @property
    def pair_losses(self) -> torch.Tensor:
        """(P,) weighted, gated, nan_to_num'ed pair losses of the last iteration_a."""
        return self._view(4, self.P)

    @property
    def flat(self) -> torch.Tensor:
        """6S + 1 floats: d loss / d (dr_s, dt_s) of every submap from THIS plan's pairs, then their loss sum."""
        return self._view(5, 6 * self.S + 1)

    @property
    def flat_reduce(self) -> torch.Tensor:
        """What a multi-rank caller sums between iteration_a and iteration_b: ``flat`` followed by S counts "pairs of
        submap s (in this plan) that passed the overlap gate" -- iteration_b leaves a submap with none alone, as
        torch.optim.Adam leaves a parameter without a gradient."""
        return self._view(5, 7 * self.S + 1)

    @property
    def adam_steps(self) -> torch.Tensor:
        """(S,) int32: Adam steps each submap's pose has taken."""
        return self._view(11, self.S).view(torch.int32)

    def iteration_a(self):
        # iteration_b leaves the next iteration's poses, their ring snapshot and cleared accumulators: no prologue launch
        self.cfg.poses_ready = 1 if getattr(self, "_poses_ready", False) else 0
        self._poses_ready = False
        _lib.check(_lib.load().miso_align_iteration_a(C.byref(self.cfg), _stream(self.state)), "miso_align_iteration_a")

    def iteration_b(self):
        _lib.check(_lib.load().miso_align_iteration_b(C.byref(self.cfg), _stream(self.state)), "miso_align_iteration_b")
        self._poses_ready = True

    def ctrl(self) -> dict:
        """Host copy of the counters (one sync): Adam steps taken, stopped flag, iterations run, NaN-skipped."""
        c = self._view(8, 8).view(torch.int32).cpu().tolist()
        return dict(steps=c[0], stopped=bool(c[1]), iterations=c[2], skipped=c[3])

    def ring(self) -> torch.Tensor:
        """(ring_iters, 2 [+ 16 S]) rows {loss, relative change [, (S,4,4) poses before the step]}."""
        row = self._off[10]
        return self._view(9, self.ring_iters * row).view(self.ring_iters, row)


def overlap_count(R_src, t_src, R_dst, t_dst, coords_src, bound_dst) -> torch.Tensor:
    """0-d device tensor: how many of coords_src (N,3) fall inside bound_dst ((3,2) [min,max] rows,
    inclusive) after src -> world -> dst (GridAtlas.check_submap_intersection,
    grid_opt/models/grid_atlas.py:405-420).  One pass, no host sync (miso_overlap_count)."""
    _require_hip(R_src, t_src, R_dst, t_dst, coords_src)
    pts = coords_src.detach().contiguous()
    n = pts.shape[0]
    pose = torch.cat((R_src.detach().reshape(9), t_src.detach().reshape(3), R_dst.detach().reshape(9),
                      t_dst.detach().reshape(3))).contiguous()
    b = bound_dst.detach().cpu().tolist() if isinstance(bound_dst, torch.Tensor) else bound_dst
    bmin = (C.c_float * 3)(*[float(b[a][0]) for a in range(3)])
    bmax = (C.c_float * 3)(*[float(b[a][1]) for a in range(3)])
    out = torch.empty(1, device=pts.device, dtype=torch.float32)
    _lib.check(_lib.load().miso_overlap_count(_ptr(pose), _ptr(pts), n, bmin, bmax, _ptr(out), _stream(pts)),
               "miso_overlap_count")
    return out[0]


# --------------------------------------------------------------------------- #
# tracker: Gauss-Newton normal equations
# --------------------------------------------------------------------------- #
def lm_normal_eq(coords_frame, R_frame, grad_world, sdf_pred, sdf_gt, loss_type="L2", gm_scale=0.1):
    """H = J^T W J (6,6), g = J^T W r (6,1), sum w r^2, for J_i = [((R x_i) x grad_i)^T R, grad_i^T]
    (Tracker.lm_step, grid_opt/slam/tracker.py:148-212) in one launch (miso_lm_normal_eq)."""
    _require_hip(coords_frame, R_frame, grad_world, sdf_pred, sdf_gt)
    x = coords_frame.detach().contiguous()
    gw = grad_world.detach().contiguous()
    Rm = R_frame.detach().contiguous()
    s = sdf_pred.detach().reshape(-1).contiguous()
    t = sdf_gt.detach().reshape(-1).contiguous()
    n = x.shape[0]
    assert x.shape == (n, 3) and gw.shape == (n, 3) and Rm.shape == (3, 3) and s.numel() == n and t.numel() == n
    lt = {"L2": 2, "GM": 3}[loss_type]
    out = torch.empty(32, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().miso_lm_normal_eq(_ptr(x), _ptr(Rm), _ptr(gw), _ptr(s), _ptr(t), n, lt, float(gm_scale),
                                             _ptr(out), _stream(x)), "miso_lm_normal_eq")
    iu = torch.triu_indices(6, 6, device=x.device)
    H = torch.zeros(6, 6, device=x.device, dtype=torch.float32)
    H[iu[0], iu[1]] = out[:21]
    H = H + H.triu(1).T
    return H, out[21:27].reshape(6, 1), out[27]


class LmTrackStep:
    """One Levenberg-Marquardt step of the keyframe tracker as ONE library call and ONE host synchronisation
    (miso_lm_track_step): pose from the corrections, samples into the submap frame, fused SDF forward + coordinate
    backward, normal equations, damped solve, corrections updated in place; the truncation filter, the frame-id /
    validity checks and the field-of-view count that the reference settles with host round trips come back as
    counters.  Scratch is sized for ``n`` rows and reused."""

    def __init__(self, n: int, device, pack: DecoderPack):
        self.n = int(n)
        f32 = dict(device=device, dtype=torch.float32)
        mw = sdf_mask_words(pack)
        self.pose = torch.empty(12, **f32)
        self.xw = torch.empty((self.n, 3), **f32)
        self.sdf = torch.empty(self.n, **f32)
        self.grad = torch.empty((self.n, 3), **f32)
        self.ones = torch.ones(self.n, **f32)
        self.mask = torch.empty(((self.n + 63) // 64) * 64 * mw, device=device, dtype=torch.int32)
        self.sums = torch.empty(36, **f32)
        self.info = torch.empty(8, **f32)
        self.clean = torch.empty(5 * max(self.n, 1), **f32)       # nan_to_num'ed coords / target / valid (sanitize=True)
        self.info_host = torch.empty(8, dtype=torch.float32, pin_memory=True)

    def _fill(self, a, coords_frame, target, valid, frame_ids, keyframe_id, trunc_dist, R_base, t_base, rot_correction,
              trans_correction, sanitize=False):
        n = self.n
        a.sanitized = self.clean.data_ptr() if sanitize else 0

        def col(c, dtypes):
            if c is None:
                return None, 1
            if c.dim() == 2 and c.shape[1] == 1:
                c = c[:, 0]
            if not (c.dim() == 1 and c.shape[0] == n and c.is_cuda and c.dtype in dtypes):
                raise ValueError("LmTrackStep wants device columns (N,) or (N,1) of the batch's length")
            return c, (c.stride(0) if n > 1 else 1)

        cf = coords_frame.detach()
        if not (cf.shape == (n, 3) and cf.is_contiguous() and cf.dtype == torch.float32 and cf.is_cuda):
            raise ValueError("LmTrackStep: coords fp32 contiguous (N,3) on the device")
        tg, a.stride_target = col(target, (torch.float32,))
        vl, a.stride_valid = col(valid, (torch.float32, torch.bool))
        fi, a.stride_frame_ids = col(frame_ids, (torch.int64,))
        for t_ in (R_base, t_base, rot_correction, trans_correction):
            assert t_.is_cuda and t_.dtype == torch.float32 and t_.is_contiguous()
        assert R_base.numel() == 9 and t_base.numel() == 3 and rot_correction.numel() == 3 and trans_correction.numel() == 3
        a.coords_frame, a.target, a.valid, a.frame_ids = cf.data_ptr(), tg.data_ptr(), 0 if vl is None else vl.data_ptr(), \
            0 if fi is None else fi.data_ptr()
        a.valid_is_bool = int(vl is not None and vl.dtype == torch.bool)
        a.n, a.keyframe_id = n, int(keyframe_id)
        a.trunc_dist = -1.0 if trunc_dist is None else float(trunc_dist)
        a.R_base, a.t_base = R_base.data_ptr(), t_base.data_ptr()
        a.rot_correction, a.trans_correction = rot_correction.data_ptr(), trans_correction.data_ptr()
        a.pose, a.coords_world, a.sdf, a.grad = self.pose.data_ptr(), self.xw.data_ptr(), self.sdf.data_ptr(), self.grad.data_ptr()
        a.ones, a.relu_mask, a.sums, a.info = self.ones.data_ptr(), self.mask.data_ptr(), self.sums.data_ptr(), self.info.data_ptr()
        return cf

    def __call__(self, features, meta: GridMeta, pack: DecoderPack, coords_frame, target, valid, frame_ids, keyframe_id,
                 trunc_dist, R_base, t_base, rot_correction, trans_correction, loss_type: str, gm_scale: float,
                 lm_lambda: float, sanitize: bool = False):
        """-> [|delta_R| rad, |delta_t|, |g|, rows in bound, rows kept, wrong frame ids, invalid rows, 0] (host floats).
        rot_correction / trans_correction: 3-float views of the pose parameters, updated in place.
        sanitize: torch.nan_to_num on the batch first (prepare_batch folded in)."""
        a = _lib.LmTrack()
        cf = self._fill(a, coords_frame, target, valid, frame_ids, keyframe_id, trunc_dist, R_base, t_base, rot_correction,
                        trans_correction, sanitize)
        a.loss_type, a.gm_scale, a.lm_lambda = {"L2": 2, "GM": 3}[loss_type], float(gm_scale), float(lm_lambda)
        m, packed = pack.get()
        g = _fill_grid([f.detach() for f in features], meta)
        _lib.check(_lib.load().miso_lm_track_step(C.byref(g), C.byref(m), _ptr(packed), C.byref(a), _stream(cf)),
                   "miso_lm_track_step")
        self.info_host.copy_(self.info, non_blocking=True)
        torch.cuda.current_stream(cf.device).synchronize()
        return self.info_host.tolist()


class TrackAdamWindow(LmTrackStep):
    """The tracker's Adam solver on the device (miso_track_adam_step): every iteration is one library call and nothing
    is read back until ``finish()``.  A window = a fresh optimizer (zero moments, step count 0), like the Trainer the
    reference builds per Tracker.track_window."""

    def __init__(self, n: int, device, pack: DecoderPack, lr: float, iterations: int, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(n, device, pack)
        self.iterations = int(iterations)
        host = torch.empty((max(self.iterations, 1), 6), dtype=torch.float32)
        _lib.check(_lib.load().miso_adam_scalars_table(lr, betas[0], betas[1], eps, 1, host.shape[0],
                                                       C.c_void_p(host.data_ptr())), "miso_adam_scalars_table")
        self.table = host.to(device)
        self.hyper = (float(lr), tuple(betas), float(eps), self.iterations)
        self.state = torch.zeros(16, device=device, dtype=torch.float32)
        self.ring = torch.zeros(max(self.iterations, 1), device=device, dtype=torch.float32)
        self.gpred = torch.empty(self.n, device=device, dtype=torch.float32)

    def reset(self):
        self.state.zero_()

    def step(self, features, meta: GridMeta, pack: DecoderPack, coords_frame, target, valid, frame_ids, keyframe_id,
             trunc_dist, R_base, t_base, rot_correction, trans_correction, loss_type: str, weight_sdf: float,
             gm_scale: float, sanitize: bool = False):
        t = _lib.TrackAdam()
        cf = self._fill(t.s, coords_frame, target, valid, frame_ids, keyframe_id, trunc_dist, R_base, t_base,
                        rot_correction, trans_correction, sanitize)
        t.loss_type, t.weight_sdf, t.gm_scale = {"L1": 1, "L2": 2, "GM": 3}[loss_type], float(weight_sdf), float(gm_scale)
        t.grad_pred, t.adam_table, t.adam_table_len = self.gpred.data_ptr(), self.table.data_ptr(), self.table.shape[0]
        t.state, t.loss_ring, t.ring_len = self.state.data_ptr(), self.ring.data_ptr(), self.ring.shape[0]
        m, packed = pack.get()
        g = _fill_grid([f.detach() for f in features], meta)
        _lib.check(_lib.load().miso_track_adam_step(C.byref(g), C.byref(m), _ptr(packed), C.byref(t), _stream(cf)),
                   "miso_track_adam_step")

    def finish(self):
        """One synchronisation: (losses of the iterations run, Adam steps taken, steps skipped for a NaN loss)."""
        cnt = self.state[12:15].view(torch.int32).cpu().tolist()
        return self.ring[:cnt[2]].cpu().tolist(), cnt[0], cnt[1]


# --------------------------------------------------------------------------- #
# sample generation: posed depth frames -> SDF training rows
# --------------------------------------------------------------------------- #
class RayBatch:
    """Device buffers of one ray batch (capacity n_rays * S rows) plus the two live counters.

    ``aux`` columns are {sdf, valid, sign, weight}: the table MappingStep reads, so a trainer can hand
    ``coords_frame`` / ``aux`` on without a copy.  ``rows()`` reads the live count back (one sync)."""

    def __init__(self, n_rays: int, samples_per_ray: int, device, keep_world=False):
        cap = n_rays * samples_per_ray
        self.n_rays, self.S, self.capacity = n_rays, samples_per_ray, cap
        self.coords_frame = torch.empty(cap, 3, device=device, dtype=torch.float32)
        self.sample_frame_ids = torch.empty(cap, device=device, dtype=torch.int64)
        self.aux = torch.empty(cap, 4, device=device, dtype=torch.float32)
        self.pc_world = torch.empty(cap, 3, device=device, dtype=torch.float32) if keep_world else None
        self.z_vals = torch.empty(cap, device=device, dtype=torch.float32) if keep_world else None
        self.counts = torch.zeros(4, device=device, dtype=torch.int32)   # rays after filter 1, rays kept, live rows, 0

    @property
    def live_rows(self) -> torch.Tensor:
        """(1,) int32 on the device: what the padded trainer step divides its means by."""
        return self.counts[2:3]

    def rows(self) -> int:
        return int(self.counts[2].item())


class RaySampler:
    """Frames + sampling knobs bound once (ctypes structs, workspace); each call is one miso_sample_rays."""

    def __init__(self, depth, T_WC, R_wk, t_wk, intrinsics, *, min_depth, dist_behind_surf, trunc_dist, n_strat,
                 n_surf, rays_per_frame=0, normals=None, frame_ids=None):
        _require_hip(depth, T_WC, R_wk, t_wk, normals)
        dev = depth.device
        B, H, W = depth.shape

        def f32(t, shape):
            t = t.detach().to(torch.float32).contiguous()
            assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
            return t

        self._keep = [f32(depth, (B, H, W)), f32(T_WC, (B, 4, 4)), f32(R_wk, (B, 3, 3)),
                      f32(t_wk.reshape(B, 3), (B, 3))]
        fr = _lib.RayFrames()
        fr.depth, fr.T_WC, fr.R_wk, fr.t_wk = (t.data_ptr() for t in self._keep)
        if normals is not None:
            self._keep.append(f32(normals, (B, H, W, 3)))
            fr.normals = self._keep[-1].data_ptr()
        if frame_ids is not None:
            self._keep.append(frame_ids.detach().to(device=dev, dtype=torch.int64).contiguous())
            assert self._keep[-1].numel() == B
            fr.frame_ids = self._keep[-1].data_ptr()
        fr.n_frames, fr.H, fr.W = B, H, W
        fr.fx, fr.fy, fr.cx, fr.cy = (float(v) for v in intrinsics)
        edges = torch.linspace(0, 1, n_strat + 1, dtype=torch.float32)        # utils_sample.py:214-216
        self._edges = (C.c_float * (n_strat + 1))(*edges.tolist())
        self.cfg = _lib.RaySampling(float(min_depth), float(dist_behind_surf), float(trunc_dist), int(n_strat),
                                    int(n_surf), int(rays_per_frame), C.cast(self._edges, C.POINTER(C.c_float)))
        self.frames, self.device, self.B = fr, dev, B
        self.n_strat, self.n_surf, self.S = int(n_strat), int(n_surf), int(n_strat + n_surf)
        self._ws = None

    def __call__(self, pix_h, pix_w, u, g, pix_b=None, out: Optional[RayBatch] = None, keep_world=False) -> RayBatch:
        _require_hip(u, g)
        n_rays = pix_h.numel()
        for t in (pix_h, pix_w, pix_b):
            if t is not None and not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == n_rays):
                raise RuntimeError("pixel indices must be contiguous int64 tensors on the HIP device, one per ray")
        if self.n_strat > 0:
            assert u.is_contiguous() and tuple(u.shape) == (n_rays, self.n_strat)
        if self.n_surf > 1:
            assert g.is_contiguous() and tuple(g.shape) == (n_rays, self.n_surf - 1)
        if out is None:
            out = RayBatch(n_rays, self.S, self.device, keep_world=keep_world)
        assert out.n_rays == n_rays and out.S == self.S
        lib = _lib.load()
        need = max(int(lib.miso_sample_rays_workspace_bytes(n_rays, self.B)), 1)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device, dtype=torch.uint8)
        _lib.check(lib.miso_sample_rays(C.byref(self.frames), C.byref(self.cfg), n_rays, _ptr(pix_b), _ptr(pix_h),
                                        _ptr(pix_w), _ptr(u if self.n_strat > 0 else None),
                                        _ptr(g if self.n_surf > 1 else None), _ptr(self._ws), _ptr(out.coords_frame),
                                        _ptr(out.sample_frame_ids), _ptr(out.aux), _ptr(out.pc_world),
                                        _ptr(out.z_vals), _ptr(out.counts), _stream(out.aux)), "miso_sample_rays")
        return out


def sample_rays(depth, T_WC, R_wk, t_wk, intrinsics, pix_h, pix_w, u, g, *, min_depth, dist_behind_surf,
                trunc_dist, n_strat, n_surf, rays_per_frame=0, pix_b=None, normals=None, frame_ids=None,
                out: Optional[RayBatch] = None, keep_world=False) -> RayBatch:
    """PosedSdfRgbd.getitem_sdf (grid_opt/datasets/sdf_rgbd.py:381-483) for one batch of pixels, on the device:
    miso_sample_rays.  ``intrinsics`` = (fx, fy, cx, cy); draws ``u`` (n_rays, n_strat) and ``g`` (n_rays, n_surf-1)
    are consumed by the rays that pass the depth filter, in order."""
    sampler = RaySampler(depth, T_WC, R_wk, t_wk, intrinsics, min_depth=min_depth, dist_behind_surf=dist_behind_surf,
                         trunc_dist=trunc_dist, n_strat=n_strat, n_surf=n_surf, rays_per_frame=rays_per_frame,
                         normals=normals, frame_ids=frame_ids)
    _require_hip(u, g)
    if not (pix_h.is_cuda and pix_w.is_cuda and (pix_b is None or pix_b.is_cuda)):
        raise RuntimeError("miso_amd ops run on the HIP device only (no CPU fallback); pixel indices are on the host")
    i64 = lambda t: None if t is None else t.detach().reshape(-1).to(torch.int64).contiguous()
    c32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    return sampler(i64(pix_h), i64(pix_w), c32(u), c32(g), pix_b=i64(pix_b), out=out, keep_world=keep_world)


def marching_cubes(vol: torch.Tensor, iso: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Triangle mesh of the level set ``vol == iso`` of a dense (nx, ny, nz) fp32 volume in HBM: vertices (V, 3)
    fp32 in index coordinates (x, y, z) and triangles (T, 3) int64, both on the device.  Replaces
    ``mcubes.marching_cubes(u, threshold)`` of extract_geometry (grid_opt/utils/utils_sdf.py:89-101): one sweep
    over the volume (miso_mc_classify: sign cases + a ballot bitmap of the crossed lattice edges), then triangles
    as vertex indices (miso_mc_emit) and vertex positions (miso_mc_vertices) -- shared vertices are numbered by
    bitmap rank, nothing is sorted.  Vertices are ordered by (x, y, axis, z) of their edge, triangles by cell
    (x-major) then table order.  One host read-back (the counts size the outputs)."""
    _require_hip(vol)
    assert vol.ndim == 3, "marching_cubes takes a (nx, ny, nz) volume"
    lib = _lib.load()
    u = vol.detach().contiguous()
    nx, ny, nz = (int(s) for s in u.shape)
    dev, st = u.device, _stream(u)
    W = lib.miso_mc_words(nx, ny, nz)
    if W < 0:
        raise RuntimeError("marching_cubes: volume too large (>= 2^31 samples) or an empty axis")
    ws = torch.empty((lib.miso_mc_workspace_bytes(nx, ny, nz) + 7) // 8, dtype=torch.int64, device=dev)
    counts = torch.empty(4 * W + 2, dtype=torch.int32, device=dev)
    _lib.check(lib.miso_mc_classify(_ptr(u), nx, ny, nz, float(iso), _ptr(ws), _ptr(counts), st), "miso_mc_classify")
    offs = torch.empty(4 * W + 2, dtype=torch.int64, device=dev)
    torch.cumsum(counts[:3 * W], 0, dtype=torch.int64, out=offs[:3 * W])
    torch.cumsum(counts[3 * W:4 * W], 0, dtype=torch.int64, out=offs[3 * W:4 * W])
    offs[4 * W:] = counts[4 * W:]
    n_vert, n_tri, tri_chunks, vert_chunks = (int(v) for v in offs[[3 * W - 1, 4 * W - 1, 4 * W, 4 * W + 1]].tolist())
    offs -= counts                                   # inclusive -> exclusive
    verts = torch.empty((n_vert, 3), dtype=torch.float32, device=dev)
    faces = torch.empty((n_tri, 3), dtype=torch.int64, device=dev)
    _lib.check(lib.miso_mc_emit(nx, ny, nz, _ptr(ws), _ptr(offs), tri_chunks, n_tri, _ptr(faces), st), "miso_mc_emit")
    _lib.check(lib.miso_mc_vertices(_ptr(u), nx, ny, nz, float(iso), _ptr(ws), _ptr(offs), vert_chunks, n_vert,
                                    _ptr(verts), st), "miso_mc_vertices")
    return verts, faces


def rigid_by_index(R: torch.Tensor, t: Optional[torch.Tensor], idx: torch.Tensor, x: torch.Tensor,
                   transpose: bool = False) -> torch.Tensor:
    """y[i] = R[idx[i]] x[i] + t[idx[i]] (or R[idx[i]]^T x[i] with ``transpose``) in one pass (miso_rigid_by_index);
    R (K,3,3), t (K,3) or None, idx (N,) int64, x (N,3).  No autograd here: grid_opt.loss._RigidByIndex wraps it."""
    _require_hip(R, t, x)
    R = R.detach().contiguous()
    t = None if t is None else t.detach().contiguous()
    x = x.detach().contiguous()
    idx = idx.contiguous()
    assert idx.dtype == torch.int64 and idx.is_cuda and idx.shape == (x.shape[0],) and x.shape[1] == 3
    y = torch.empty_like(x)
    _lib.check(_lib.load().miso_rigid_by_index(_ptr(R), _ptr(t), _ptr(idx), _ptr(x), x.shape[0], R.shape[0],
                                               1 if transpose else 0, _ptr(y), _stream(x)), "miso_rigid_by_index")
    return y
