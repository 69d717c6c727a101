"""DenseAdam's NaN-guard bookkeeping with the loss handed over through a host slot (ops.HostTotal), on the host alone: a
stand-in slot that becomes ready on demand.  (The kernel side -- miso_loss_total_bump_host writing the pinned ring -- is
tests/test_hip_parity.py::test_adam_device_step_host_ring_and_multi_tensor_launch.)"""
import math

import torch

from miso_amd.optim import DenseAdam


class Slot:
    def __init__(self, value, ready=False):
        self._value, self._ready, self.polls = value, ready, 0

    def ready(self):
        self.polls += 1
        return self._ready

    def value(self):
        assert self._ready
        return self._value


def _opt():
    p = torch.nn.Parameter(torch.zeros(4))
    opt = DenseAdam([p], lr=1e-3)
    st = {"step": 0}
    return opt, st


def test_guards_resolve_in_order_and_take_the_count_back_for_nan():
    opt, st = _opt()
    guard = torch.tensor(0.0)
    slots = [Slot(0.5), Slot(float("nan")), Slot(0.25)]
    for s in slots:
        st["step"] += 1
        assert opt.note_guarded_step(guard, [st], host=s) == 0
    # nothing has arrived: a non-blocking look changes nothing and stops at the first slot
    assert opt.resolve_guard(block=False) == 0 and st["step"] == 3 and slots[1].polls == 0
    # the SECOND arrives before the first: still nothing (steps are accounted for in order)
    slots[1]._ready = True
    assert opt.resolve_guard(block=False) == 0 and st["step"] == 3
    slots[0]._ready = True
    assert opt.resolve_guard(block=False) == 1 and st["step"] == 2          # the NaN step is taken back
    assert opt.skipped_steps == 1 and len(opt._pending) == 1
    slots[2]._ready = True
    assert opt.resolve_guard() == 0 and not opt._pending and st["step"] == 2


def test_a_full_queue_waits_for_the_oldest_guard_only():
    opt, st = _opt()
    guard = torch.tensor(0.0)
    slots = [Slot(float("nan") if i == 0 else 1.0) for i in range(9)]
    for s in slots[:8]:
        st["step"] += 1
        opt.note_guarded_step(guard, [st], host=s)
    assert len(opt._pending) == 8
    slots[0]._ready = True               # (a blocking wait polls until the slot is ready: make it so beforehand)
    st["step"] += 1
    skipped = opt.note_guarded_step(guard, [st], host=slots[8])
    # room was made by resolving slot 0 (NaN: reported to the caller, count taken back); slots 1.. were looked at without
    # waiting and are still pending together with the new one
    assert skipped == 1 and st["step"] == 8 and len(opt._pending) == 8
    assert slots[1].polls >= 1 and all(s.polls == 0 for s in slots[2:])
    for s in slots[1:]:
        s._ready = True
    assert opt.resolve_guard() == 0 and not opt._pending
    assert math.isnan(slots[0].value())
