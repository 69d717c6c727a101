"""Dense Adam as a torch.optim.Optimizer backed by miso_adam_active (miso_adam_dense's results without the
passes over elements that no gradient has ever reached: their moments are zero and their update is exactly zero).

Same update and state layout ('step', 'exp_avg', 'exp_avg_sq') as
torch.optim.Adam(amsgrad=False, weight_decay=0), which is what the reference's
trainers and alignment loops construct (grid_opt/trainer.py:96-97, :424-437;
grid_opt/align/base.py:110-111).  Big dense tensors (the feature grids) take one
HIP launch each; tiny ones (pose vectors) and CPU tensors use the identical
formula in a few torch ops.  Parameters whose .grad is None are skipped, as torch does.

NaN guard without a host round trip: ``step(guard=loss)`` hands the device scalar to the kernels, which leave
everything alone if it is NaN (what the reference does after reading the loss back, grid_opt/trainer.py:213-219).
The host learns about a skipped step one call later (a pinned copy + event, resolved before the next step is
launched, by which time it has long completed) and takes the step count back, so the bias corrections stay those
of torch.optim.Adam."""
import logging
import math
import time

import torch

from . import ops

logger = logging.getLogger(__name__)

_KERNEL_MIN_NUMEL = 4096


class DenseAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # the remaining torch.optim.Adam hyper-parameters at their defaults, so that a saved state dict loads into
        # torch.optim.Adam (its step() reads them) and the other way round
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self.skipped_steps = 0
        self._pending = []            # guarded steps not yet accounted for: [pinned loss copy, event, states, device scalar]
        self._guard_slots = []        # pinned floats + events, reused

    def resolve_guard(self, block=True, at_most=None):
        """Account for guarded steps whose loss has arrived: if it was NaN the device skipped the step -- take the step
        counts back.  block=True waits for every pending one (the launch-by-launch path computes the next step's bias
        corrections from the count, so it must be right); block=False looks only at those that have completed (a
        captured step counts on the device).  at_most: wait for that many of the oldest ones only, then go on without
        waiting.  Returns the number of steps found skipped."""
        pending = self.__dict__.get('_pending')
        skipped = 0
        while pending:
            if at_most is not None:
                if at_most <= 0:
                    block = False
                at_most -= 1
            host, event, states, src, own = pending[0]
            if own:
                if block:
                    event.synchronize()
                elif not event.query():
                    break
                bad = bool(torch.isnan(host[0]))
                self.__dict__.setdefault('_guard_slots', []).append((host, event))
            elif host.ready():                     # ops.HostTotal: the step's own launch left the loss in pinned memory
                bad = math.isnan(host.value())
            elif not block:
                break
            else:
                # wait for THIS step only: poll its slot (anything put on the stream -- an .item() of the device scalar --
                # would queue up behind every step launched since and drain them all)
                t0 = time.perf_counter()
                while not host.ready() and time.perf_counter() - t0 < 0.5:
                    pass
                if host.ready():
                    bad = math.isnan(host.value())
                else:
                    # half a second is thousands of steps: the host's launch count and the device's disagree (somebody
                    # launched the step without MappingStep.run).  The device scalar has the same number.
                    if not self.__dict__.get('_ring_warned'):
                        self.__dict__['_ring_warned'] = True
                        logger.warning("the loss had not arrived in the host ring after 0.5 s (a busy GPU is enough): waiting on the device scalar instead")
                    bad = bool(torch.isnan(src).item())
            pending.pop(0)
            if bad:
                for st in states:
                    st['step'] -= 1
                skipped += 1
                logger.warning("Loss is nan! Skip backward step.")
        self.skipped_steps = self.__dict__.get('skipped_steps', 0) + skipped
        return skipped

    def state_dict(self):
        self.resolve_guard()
        sd = super().state_dict()
        # 'active' is derived (rebuilt on load) and left out so that the file is a plain Adam state dict.  The
        # per-parameter dicts super() returns ARE the live ones (sd['state'][k] is self.state[p]): build new dicts,
        # never pop from them -- a checkpoint in the middle of training must not take the flags away from the step.
        sd['state'] = {k: {kk: vv for kk, vv in st.items() if kk != 'active'} for k, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        """Accepts the state of torch.optim.Adam (reference checkpoints, grid_opt/trainer.py:319-329): its 'step' is a
        float tensor, the kernels take an int; the 'active' chunk flags are rebuilt on the next step."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            if group.get('weight_decay', 0) or group.get('amsgrad', False) or group.get('maximize', False):
                raise ValueError("DenseAdam implements Adam(weight_decay=0, amsgrad=False, maximize=False) only")
        for st in self.state.values():
            if 'step' in st:
                st['step'] = int(st['step'].item()) if torch.is_tensor(st['step']) else int(st['step'])
            st.pop('active', None)

    @torch.no_grad()
    def step(self, closure=None, clear_grads=False, guard=None, touched=None):
        """clear_grads: leave every .grad zeroed (in the same pass for the big tensors), for callers that
        accumulate into persistent gradient buffers and would otherwise memset them before the next backward.
        guard: device scalar; NaN => nothing is updated (see the module docstring).
        touched: {id(param): uint8 flags} left by the scatter kernels for that parameter's .grad
        (MappingStep.touched): the step finds the chunks to move from the flags instead of reading the gradient.
        Only for gradients written by nothing but those kernels since the last step."""
        self.resolve_guard()
        if guard is not None and not guard.is_cuda:
            if bool(torch.isnan(guard)):                  # host tensors: the plain check
                logger.warning("Loss is nan! Skip backward step.")
                self.skipped_steps += 1
                if clear_grads:
                    for group in self.param_groups:
                        for p in group['params']:
                            if p.grad is not None:
                                p.grad.zero_()
                return None
            guard = None
        ok = None                     # device flag "loss is a number", formed only if a tiny tensor needs it
        stepped = []
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (b1, b2), eps = group['lr'], group['betas'], group['eps']
            batch = []
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                stepped.append(st)
                t, m, v = st['step'], st['exp_avg'], st['exp_avg_sq']
                if (p.is_cuda and p.dtype == torch.float32 and p.numel() >= _KERNEL_MIN_NUMEL
                        and g.stride() == p.stride() and m.stride() == p.stride()):
                    if 'active' not in st:
                        # state restored from a plain Adam checkpoint: everything may already be moving
                        st['active'] = ops.adam_active_flags(p)
                        if t > 1:
                            st['active'].fill_(1)
                    tch = None if touched is None else touched.get(id(p))
                    if tch is None:
                        batch.append((p, g, m, v, st['active'], t))      # launched together below
                    else:
                        ops.adam_active_(p, g, m, v, st['active'], t, lr, b1, b2, eps, zero_grad=clear_grads,
                                         guard=None if guard is None else guard.detach().reshape(1), touched=tch)
                    # the kernel wrote through raw pointers: tell autograd / version-keyed caches (DecoderPack)
                    torch.autograd.graph.increment_version(p)
                    continue
                if guard is not None:
                    if ok is None:
                        ok = ~torch.isnan(guard.detach()).reshape(())
                    keep = (p.clone(), m.clone(), v.clone())
                m.lerp_(g, 1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(m, denom, value=-(lr / bc1))
                if guard is not None:                     # tiny tensors: undo on the device if the loss was NaN
                    for cur, old in zip((p, m, v), keep):
                        cur.copy_(torch.where(ok, cur, old))
                if clear_grads:
                    g.zero_()
            self._launch_batch(batch, lr, b1, b2, eps, clear_grads, None if guard is None else guard.detach().reshape(1))
        if guard is not None and stepped:
            self.note_guarded_step(guard, stepped)
        return loss

    def _launch_batch(self, batch, lr, b1, b2, eps, clear_grads, guard):
        """The kernel-sized tensors of one parameter group: ONE launch when they are at the same step (the levels of a
        grid are; miso_adam_active_multi), else one each.  The argument block is kept while the addresses stay."""
        if not batch:
            return
        same_t = all(b[5] == batch[0][5] for b in batch)
        if len(batch) < 2 or len(batch) > ops._lib.ADAM_MAX_TENSORS or not same_t:
            for p, g, m, v, act, t in batch:
                ops.adam_active_(p, g, m, v, act, t, lr, b1, b2, eps, zero_grad=clear_grads, guard=guard)
            return
        key = tuple(x.data_ptr() for b in batch for x in b[:5]) + (bool(clear_grads),)
        cache = self.__dict__.get('_multi_block')
        if cache is None or cache[0] != key:
            cache = self.__dict__['_multi_block'] = (key, ops.adam_tensors([b[:5] + (clear_grads,) for b in batch]))
        ops.adam_active_multi_(cache[1], batch[0][5], lr, b1, b2, eps, guard=guard)

    def note_guarded_step(self, guard, stepped, host=None):
        """A step guarded by the device scalar ``guard`` has been launched for the states ``stepped`` (their 'step'
        already counts it): remember to take the count back if the guard turns out NaN (resolve_guard).
        Returns the number of EARLIER steps found skipped while making room (a bounded number of guards is kept in
        flight): a caller that mirrors a device-side step counter subtracts it from its mirror -- the device counter
        never moved for those steps, and overwriting it with the host count while the step just launched is still
        unresolved could leave it one ahead.
        host: the ops.HostTotal of a step whose own launch hands the guard value to the host (AdamDeviceStep.ring) --
        then nothing is put on the stream, neither a copy nor an event (an event record is a barrier packet: 5.6 us
        between the step's last kernel and the next step's first)."""
        pending = self.__dict__.setdefault('_pending', [])        # (an unpickled optimizer has no such attributes)
        slots = self.__dict__.setdefault('_guard_slots', [])
        skipped = 0
        if len(pending) >= 8:
            # a bounded number of guards in flight: wait for the oldest only -- waiting for all of them would drain the
            # queue the host has filled and leave the device idle until the next step is launched
            skipped = self.resolve_guard(block=True, at_most=len(pending) - 7)
        src = guard.detach().reshape(1)
        own = host is None
        event = None
        if own:
            host, event = slots.pop() if slots else (torch.empty(1, dtype=torch.float32, pin_memory=True), torch.cuda.Event())
            host.copy_(src, non_blocking=True)
            event.record()
        # the device scalar stays referenced until the copy has been consumed: a caller that drops the loss
        # right away would hand its block back to the allocator while the asynchronous copy may still read it
        pending.append((host, event, stepped, src, own))
        return skipped
