"""One mapping iteration of a submap with a frozen decoder as a fixed launch
sequence.  Large batches (binned): sort -> fused encode+decode forward -> fused backward with the
mapping loss folded in -> owner-computes grid gradient [-> dense Adam].  Small batches: clear
grads -> forward -> mapping loss (+ d/d pred) -> backward with atomic scatter [-> Adam].

This is the trainer step of grid_opt/trainer.py:196-228 for MisoLossMapping
(grid_opt/loss.py:754-813, pose variables locked as in Mapper.mapping,
grid_opt/slam/mapper.py:72-75) without the per-op launches, host syncs and
autograd bookkeeping; the sequence is capturable in a HIP graph.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import dataclasses
import os

import torch

from . import ops


class MappingStep:
    TOUCHED_MIN_NUMEL = 32 << 20      # floats: levels from 128 MB up keep touched-chunk flags for Adam
    CROWDED_MIN_POINTS = 16384      # crowded=True: batches are binned from this size (default SortedBatch.AUTO_MIN_POINTS)
    STREAM_MIN_POINTS = 131072      # use_graph=None: batches from this size run as plain stream launches

    def __init__(self, features: Sequence[torch.Tensor], meta: ops.GridMeta, pack: ops.DecoderPack,
                 n_points: int, loss_type: str = "L1", weight_sdf: float = 1.0, weight_fs: float = 0.0,
                 trunc_dist: float = 0.0, adam: Optional[dict] = None, use_graph: Optional[bool] = None,
                 sort: Optional[bool] = None, need_levels: Optional[Sequence[bool]] = None,
                 keep_sdf: bool = True, padded: bool = False, grads_cleared_by_optimizer: bool = False,
                 share_grads: Optional[Sequence[Optional[torch.Tensor]]] = None,
                 adam_device: Optional["ops.AdamDeviceStep"] = None,
                 adam_state: Optional[Sequence[Optional[tuple]]] = None, crowded: bool = False):
        """crowded: the batches are ray samples (they pile up around surfaces and cameras instead of filling the
        bound): bin from CROWDED_MIN_POINTS and push the coarse levels through the matrix cores whatever the average
        density (MISO_F_CROWDED) -- on the 54 000-sample batches of tools/demo_synthetic.py the contended atomics of
        the unbinned backward cost 200 us a step, the binned step with the push 96 us.
        use_graph: True = one graph replay per step, False = plain stream launches, None = by batch size (see below).
        padded: the batch buffers hold ``n_points`` rows of which only ``self.live_rows`` (one int32 on the
        device, set through set_batch) are live, the rest neutral padding (valid = sign = weight = 0); the loss
        means divide by the live count.  Lets a sampler with a data-dependent row count (depth holes) feed ONE
        captured graph.
        grads_cleared_by_optimizer: the caller's optimizer leaves the gradient buffers zeroed after its step
        (DenseAdam.step(clear_grads=True)), so the small-batch path does not memset them before scattering.
        adam_device + adam_state: the optimizer step INSIDE the captured sequence -- per level (exp_avg, exp_avg_sq,
        active) tensors of the caller's optimizer state (they stay the caller's: checkpoints, resets) and the
        device-side step scalars of ops.AdamDeviceStep; ``self.total`` (0-d) then holds the step's loss, which also
        guards the step against NaN.  The caller mirrors the step count (adam_device.count / its state['step']).
        need_levels: which levels get a gradient (default all) -- the coarse-to-fine schedule of
        GridTrainer optimises one level at a time.  keep_sdf: also leave the predicted SDF of the batch
        in ``self.sdf`` (caller order); a training loop only needs the loss and the gradients, and on
        the binned path the scattered write costs 1.6 us."""
        self.keep_sdf = bool(keep_sdf)
        self.external_clear = bool(grads_cleared_by_optimizer)
        self.features = list(features)
        if crowded and not meta.flags & ops._lib.F_CROWDED:
            meta = dataclasses.replace(meta, flags=meta.flags | ops._lib.F_CROWDED)
        self.meta, self.pack = meta, pack
        self.n = int(n_points)
        self.loss_cfg = (loss_type, float(weight_sdf), float(weight_fs), float(trunc_dist))
        dev = self.features[0].device
        if not ops.sdf_fused_supported(self.features, meta, pack):
            raise RuntimeError("MappingStep needs a (grid, decoder) shape covered by the fused kernels")
        f32 = dict(device=dev, dtype=torch.float32)
        # static buffers (graph-safe): inputs are copied into these
        self.x = torch.zeros((self.n, 3), **f32)
        # loss inputs {target, valid, sign, weight}: one (N,4) row per point (the binned forward reads
        # a point's loss inputs with one 16-B load); the four columns are views
        self.aux = torch.zeros((self.n, 4), **f32)
        self.aux[:, 1] = 1.0
        self.aux[:, 3] = 1.0
        self.target, self.valid = self.aux[:, 0:1], self.aux[:, 1:2]
        self.sign, self.weight = self.aux[:, 2:3], self.aux[:, 3:4]
        self.sdf = torch.empty((self.n, 1), **f32)
        self.gpred = torch.empty((self.n, 1), **f32)
        self.loss_slots = torch.zeros((ops._lib.LOSS_SLOTS, 2), **f32)   # binned path: per-workgroup sums
        self._loss = torch.zeros(2, **f32)
        need = [True] * len(self.features) if need_levels is None else [bool(v) for v in need_levels]
        assert len(need) == len(self.features)
        self.need_levels = need
        # share_grads: gradient buffers of a previous step over the same grids (a trainer whose batch size changes
        # builds a step per size; the dense buffers -- the size of the grids -- are not re-allocated each time)
        if share_grads is not None and all((g is not None) == nd and (g is None or g.shape == f.shape)
                                           for g, f, nd in zip(share_grads, self.features, need)):
            self.grads = list(share_grads)
            self._shared_grads = True
        else:
            self.grads = [torch.zeros_like(f) if nd else None for f, nd in zip(self.features, need)]
        # one flag byte per ADAM_CHUNK (64) gradient floats, set by the scatter kernels where they put a non-zero: Adam finds the
        # chunks a batch wrote from these instead of reading the whole gradient (a 144 M-float Newer College level:
        # 0.6 MB of flags instead of 576 MB: 124 -> 28 us with 2 % of the chunks moving).  Valid because nothing but
        # this step's kernels writes self.grads.  Only for big levels: where every chunk moves anyway the flag-driven
        # launch is ~5 % slower than the scan (tools/adam_bench.py), and a level sized to the scene ends up there.
        self.touched = [ops.adam_active_flags(f) if nd and f.numel() >= self.TOUCHED_MIN_NUMEL else None
                        for f, nd in zip(self.features, need)]
        self.adam = adam
        if adam is not None:
            self.exp_avg = [torch.zeros_like(f) if nd else None for f, nd in zip(self.features, need)]
            self.exp_avg_sq = [torch.zeros_like(f) if nd else None for f, nd in zip(self.features, need)]
            self.active = [ops.adam_active_flags(f) if nd else None for f, nd in zip(self.features, need)]
            self.t = 0
        self.live_rows = torch.full((1,), self.n, device=dev, dtype=torch.int32) if padded else None
        if padded:
            sort = True        # the live count is read by the binned forward
        if sort is None:   # default: bin when the batch is large enough for it to pay
            floor = self.CROWDED_MIN_POINTS if self.meta.flags & ops._lib.F_CROWDED else ops.SortedBatch.AUTO_MIN_POINTS
            sort = ops.SortedBatch.AUTO_MIN_POINTS is not None and self.n >= floor
        # Binning: 16 tiles per axis.  MISO_STEP_TILES=auto (dev / tests) lets it follow the grids -- more tiles on an axis
        # where the finest level needs them to be owned by the pull (ops.choose_tiles: (25, 16, 25) for a ScanNet submap,
        # every level then goes through the matrix-core pull).  Measured at the ScanNet shape (540 000 samples on 18 % of
        # the bound): 412 us per trainer step against 324 us with the fine level scattered from the train kernel and the
        # coarse one pushed -- a crowded block works its table off in epochs on ONE workgroup while most of the chip has
        # no block to work on (tools/experiments/README.md) -- so it is not the default.
        import os
        env = os.environ.get("MISO_STEP_TILES")
        self.tiles = ops.choose_tiles(self.features) if env == "auto" else ops.pack_tiles(int(env or ops.SortedBatch.TILES))
        # the fused step (sort -> sdf_train_kernel -> pull) reads a point's original index out of xn[p].w: no perm[] array,
        # one scattered store per point less in the sort (sort_scatter_kernel 11.2 -> 8.6 us at 262 144 points)
        self.sorted = ops.SortedBatch(self.n, dev, tiles=self.tiles, need_perm=not self._fused_train()) if sort else None
        if getattr(self, "_shared_grads", False) and self.sorted is None:
            # the small-batch path accumulates onto buffers it expects zeroed; the previous owner may have been a
            # binned step, which overwrites and never clears
            for g in self.grads:
                if g is not None:
                    g.zero_()
        self.adam_device, self.adam_state = adam_device, adam_state
        if adam_device is not None:
            assert adam is None and adam_state is not None and len(adam_state) == len(self.features)
            assert all((st is not None) == nd for st, nd in zip(adam_state, need))
        self.total = torch.zeros((), **f32)
        # binned step with the optimizer inside: the levels the backward ADDS to (atomic scatter, push) are cleared by
        # the Adam launch that consumes them instead of by a fill in front of every backward (64 MB at the ScanNet shape)
        self._adam_clears = 0
        if adam_device is not None and self.sorted is not None:
            self._adam_clears = ops.sdf_bwd_scattered_levels(self.features, meta, self.grads, self.n, tiles=self.tiles)
            for l, g in enumerate(self.grads):
                if g is not None and (self._adam_clears >> l) & 1:
                    g.zero_()
        self._graph = None
        if use_graph is None:
            # a graph replay saves the host's launch work (decisive for small batches: the step is launch-bound) and
            # costs ~6 us of device idle time between two replays (measured on the cfg-2 step: 163.3 us per replay,
            # 157.0 us as plain stream launches with the host 35 us per step, tools/graph_vs_eager.py) -- from
            # STREAM_MIN_POINTS the device time per step is several times the host's and plain launches win.  (The
            # host may run ahead as far as the runtime's queue lets it, ~47 steps; what does starve the device is a
            # full collection of Python's garbage collector, ~45 ms over the objects `import torch` leaves behind:
            # a long-running loop calls gc.freeze() after its setup, as bench.py does.)
            use_graph = self.n < self.STREAM_MIN_POINTS
        self._use_graph = use_graph and adam is None  # the Adam step count changes per call (adam_device: on the device)

    def set_batch(self, x, target, valid=None, sign=None, weight=None, live_rows=None):
        if self.live_rows is not None:
            assert live_rows is not None, "a padded step needs the live row count of every batch"
            self.live_rows.copy_(live_rows.reshape(1))
        self.x.copy_(x.reshape(self.n, 3))
        cols = (target, valid, sign, weight)
        if all(c is not None and c.dtype == torch.float32 and c.device == self.aux.device for c in cols):
            # the four label columns interleaved into the (N,4) rows by ONE launch instead of four strided copies
            torch.cat([c.reshape(self.n, 1) for c in cols], dim=1, out=self.aux)
            return
        self.target.copy_(target.reshape(self.n, 1))
        for buf, src, fill in ((self.valid, valid, 1.0), (self.sign, sign, 0.0), (self.weight, weight, 1.0)):
            if src is None:
                buf.fill_(fill)
            else:
                buf.copy_(src.reshape(self.n, 1))

    def _launch(self):
        lt, ws, wf, td = self.loss_cfg
        # binned path: gradients are written owner-computes, nothing to clear
        need_zero = self.sorted is None and (self.adam is None or self.t == 0) and not self.external_clear
        if need_zero:
            for g in self.grads:
                if g is not None:
                    g.zero_()
        L = len(self.features)
        if self.sorted is not None and self._fused_train():
            # binned path: forward + loss + decoder backward are ONE launch (sdf_train_kernel, which also scatters the
            # levels whose bricks are beyond the pull's reach), then the pull / push of the other levels
            self.sorted.sort(self.x, self.meta)   # part of the step: a new batch arrives every iteration
            ops.sdf_train_raw(self.features, self.meta, self.pack, self.sorted, self.aux, self.loss_slots, self.grads,
                              lt, ws, wf, td, sdf_out=self.sdf if self.keep_sdf else None, n_live=self.live_rows,
                              touched=self.touched, zeroed=self.adam_device is not None)
        elif self.sorted is not None:
            # binned path, a level still scattered from the backward kernel (bricks beyond what the pull owns)
            self.sorted.sort(self.x, self.meta)
            if getattr(self, "_mask", None) is None:
                mw = ops.sdf_mask_words(self.pack)
                self._mask = torch.empty(((self.n + 63) // 64) * 64 * mw, device=self.x.device, dtype=torch.int32)
            # forward + mapping loss: sdf, ReLU bits, d loss / d sdf (binned order) in one launch
            ops.sdf_fwd_loss_raw(self.features, self.meta, self.pack, self.sorted, self.aux, self._mask,
                                 self.gpred, self.loss_slots, lt, ws, wf, td,
                                 sdf_out=self.sdf if self.keep_sdf else None, n_live=self.live_rows)
            ops.sdf_bwd_raw(self.x, self.features, self.meta, self.pack, self.gpred, self._mask, False,
                            self.need_levels, self.grads, sorted_batch=self.sorted, overwrite=True, gsdf_sorted=True,
                            touched=self.touched, zeroed=self.adam_device is not None)
        elif self._fused_train() and self.live_rows is None:
            # unbinned (small) batch: forward + loss + decoder backward + the atomic scatter of every level, one launch
            ops.sdf_train_unsorted_raw(self.x, self.features, self.meta, self.pack, self.aux, self.loss_slots, self.grads,
                                       lt, ws, wf, td, sdf_out=self.sdf if self.keep_sdf else None, touched=self.touched)
        else:
            # forward + mapping loss in one launch here too (the label rows are read by the forward itself)
            if getattr(self, "_mask", None) is None:
                mw = ops.sdf_mask_words(self.pack)
                self._mask = torch.empty(((self.n + 63) // 64) * 64 * mw, device=self.x.device, dtype=torch.int32)
            mask = self._mask
            ops.sdf_fwd_loss_unsorted_raw(self.x, self.features, self.meta, self.pack, self.aux, mask, self.gpred,
                                          self.loss_slots, lt, ws, wf, td, sdf_out=self.sdf if self.keep_sdf else None)
            ops.sdf_bwd_raw(self.x, self.features, self.meta, self.pack, self.gpred, mask, False,
                            self.need_levels, self.grads, touched=self.touched)
        if self.adam_device is not None:
            self.adam_device.total_and_bump(self.loss_slots, self.total)      # loss sum + step count: one launch
            # all levels share ONE launch (each stepped by its gradient or by its `touched` flags): as launches of their own the coarse
            # levels of a pyramid are all ramp and tail (cfg-2: 8.5 + 13.4 us for 67 MB next to 74.5 us for 469 MB)
            # (the block bakes raw pointers in: keyed by them, so a level whose storage was rebound -- f.data = ..., a replaced
            # gradient or moment buffer -- gets a new block instead of an update of stale memory)
            akey = tuple(t.data_ptr() for p, g, st, tch in zip(self.features, self.grads, self.adam_state, self.touched)
                         if g is not None for t in (p, g, st[0], st[1], st[2]))
            multi = self.__dict__.get("_adam_multi")
            if multi is None or self.__dict__.get("_adam_multi_key") != akey:
                dense = [(p, g, st[0], st[1], st[2], self.sorted is None or bool((self._adam_clears >> l) & 1), tch)
                         for l, (p, g, st, tch) in enumerate(zip(self.features, self.grads, self.adam_state, self.touched))
                         if g is not None]
                multi = self._adam_multi = (self.adam_device.multi(dense) if 2 <= len(dense) <= ops._lib.ADAM_MAX_TENSORS
                                            and os.environ.get("MISO_ADAM_PER_LEVEL") is None else False)
                self._adam_multi_key = akey
            if multi:
                self.adam_device.step_multi_(multi, guard=self.total)
            for l, (p, g, st, tch) in enumerate(zip(self.features, self.grads, self.adam_state, self.touched)):
                if g is None or multi:
                    continue
                self.adam_device.step_(p, g, st[0], st[1], st[2], touched=tch, guard=self.total,
                                       zero_grad=self.sorted is None or bool((self._adam_clears >> l) & 1))
        if self.adam is not None:
            self.t += 1
            akey = tuple(t.data_ptr() for p, g, m, v, act in zip(self.features, self.grads, self.exp_avg, self.exp_avg_sq,
                                                               self.active) if g is not None for t in (p, g, m, v, act))
            multi = self.__dict__.get("_adam_multi")
            if multi is None or self.__dict__.get("_adam_multi_key") != akey:
                self._adam_multi_key = akey
                dense = [(p.data, g, m, v, act, self.sorted is None, tch)
                         for p, g, m, v, act, tch in zip(self.features, self.grads, self.exp_avg, self.exp_avg_sq,
                                                         self.active, self.touched) if g is not None]
                multi = self._adam_multi = (ops.adam_tensors(dense) if 2 <= len(dense) <= ops._lib.ADAM_MAX_TENSORS
                                            and os.environ.get("MISO_ADAM_PER_LEVEL") is None else False)
            if multi:
                ops.adam_active_multi_(multi, self.t, **self.adam)
            for p, g, m, v, act, tch in zip(self.features, self.grads, self.exp_avg, self.exp_avg_sq, self.active,
                                            self.touched):
                if g is None or multi:
                    continue
                # zero_grad=True: the gradient is cleared in the same pass, so the next
                # iteration needs no memset
                ops.adam_active_(p.data, g, m, v, act, self.t, zero_grad=self.sorted is None, touched=tch, **self.adam)

    def _fused_train(self) -> bool:
        """Whether the binned step runs as sort -> sdf_train_kernel -> pull (decided once: the grids do not change)."""
        ok = self.__dict__.get("_fused_train_ok")
        if ok is None:
            import os
            ok = self._fused_train_ok = (os.environ.get("MISO_NO_FUSED_TRAIN") is None      # dev: the two-launch form
                                         and ops.sdf_train_supported(self.features, self.meta, self.grads, self.pack))
        return ok

    @property
    def loss(self) -> torch.Tensor:
        """(2,) = [weight_sdf * sdf term, weight_fs * free-space term] of the last iteration."""
        return self.loss_slots.sum(dim=0)

    def run(self):
        """Launch one iteration on the current stream (asynchronous)."""
        if self.adam_device is not None:
            self.host_total = self.adam_device.note_launch()      # where the host will find this iteration's loss
        if not self._use_graph:
            self._launch()
            return
        if self._graph is None:
            self._launch()                      # this call's iteration, eagerly (allocates the buffers)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            # thread_local: another thread touching the runtime during capture (e.g. the RCCL
            # watchdog of a multi-GPU job polling events) must not invalidate it
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self._launch()                  # recorded, not executed
            return
        self._graph.replay()
