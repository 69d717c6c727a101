"""Epoch loop: batch -> loss dict -> sum -> backward -> Adam, with the coarse-to-fine
level schedule of GridTrainer (reference: grid_opt/trainer.py).

Differences from the reference that do not change results: the batch goes to the
trainer's own device (the reference ignores it and uses 'cuda:0', trainer.py:204),
TensorBoard is optional, and the NaN guard reads the loss once per step."""
import logging
import os
import time

import numpy as np
import torch
import torch.optim as optim

from miso_amd.optim import DenseAdam
from .utils.utils import PerfTimer, cond_mkdir, prepare_batch

logger = logging.getLogger(__name__)

try:  # optional: only Trainer.eval writes scalars
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # pragma: no cover
    SummaryWriter = None


def _make_optimizer(name, params, lr):
    if name == 'adam':
        return DenseAdam(params, lr=lr)
    if name == 'sgd':
        return optim.SGD(params, lr=lr)
    if name == 'lbfgs':
        raise NotImplementedError("LBFGS optimizer not implemented yet.")
    raise ValueError(f"Invalid optimizer: {name}.")



class _FastMappingPlan:
    """The captured mapping step with the optimizer INSIDE the replay, and a host side reduced to what cannot be
    captured: one launch that writes the batch into the step's static buffers (keyframe lookup + frame -> world map +
    label rows: ops.mapping_batch), one graph replay (sort, forward + loss, backward, gradient, loss sum, Adam step
    count, Adam per level), the NaN-guard bookkeeping of DenseAdam.  Built by GridTrainer._captured_mapping_step once a
    batch shape has come back; every call re-checks a fingerprint of everything the capture baked in (which tensors,
    which flags, which hyper-parameters) and hands the step back to the checked path if any of it moved.
    Why: at the Newer College shape (6 144 samples, 145 M grid floats) the GPU needs ~0.1 ms per step and the host
    side of the checked path 0.22 ms.  From MappingStep.STREAM_MIN_POINTS samples the same launches go to the stream
    one by one instead (host 105 instead of 85 us per step, the device 284 instead of 293 us at cfg-2 and 342 instead
    of 357 us at the ScanNet shape: no idle time between replays)."""

    @staticmethod
    def eligible_loss(lf, model):
        from miso_amd.grid_opt.loss import MisoLossMapping
        return (type(lf).world_coords is MisoLossMapping.world_coords
                and type(lf).query_kf_pose is MisoLossMapping.query_kf_pose
                and hasattr(model, 'kf_key_index_table') and hasattr(model, 'updated_kf_poses_all'))

    @classmethod
    def build(cls, trainer, prev_step, feats, need, pack, n, padded):
        from miso_amd import ops
        from miso_amd.optim import _KERNEL_MIN_NUMEL
        from miso_amd.step import MappingStep
        opt, lf, model = trainer.optimizer, trainer.loss_func, trainer.model
        try:
            if type(opt) is not DenseAdam or not cls.eligible_loss(lf, model):
                return None
            if opt._optimizer_step_pre_hooks or opt._optimizer_step_post_hooks:
                return None           # somebody listens to optimizer.step(): keep calling it
            mine = [f for f, nd in zip(feats, need) if nd]
            groups = [g for g in opt.param_groups if any(any(p is f for f in mine) for p in g['params'])]
            hyper = {(g['lr'], tuple(g['betas']), g['eps']) for g in groups}
            if len(hyper) != 1:
                return None
            (lr, (b1, b2), eps), = hyper
            states = []
            for f in mine:
                st = opt.state.get(f)
                if (not st or 'active' not in st or f.numel() < _KERNEL_MIN_NUMEL or not f.is_cuda
                        or st['exp_avg'].stride() != f.stride() or st['exp_avg_sq'].stride() != f.stride()):
                    return None
                states.append(st)
            opt.resolve_guard()
            if len({st['step'] for st in states}) != 1:
                return None
            dev = ops.AdamDeviceStep(lr, b1, b2, eps, feats[0].device, count=states[0]['step'])
            it = iter(states)
            adam_state = [None if not nd else (lambda st: (st['exp_avg'], st['exp_avg_sq'], st['active']))(next(it))
                          for nd in need]
            lt, ws, wf, td = prev_step.loss_cfg
            step = MappingStep([f.data for f in feats], prev_step.meta, pack, n, lt, ws, wf, td, need_levels=need,
                               keep_sdf=False, padded=padded, grads_cleared_by_optimizer=True,
                               use_graph=None,      # a replay below MappingStep.STREAM_MIN_POINTS, stream launches from there

                               sort=prev_step.sorted is not None, share_grads=prev_step.grads,
                               adam_device=dev, adam_state=adam_state)
        except (ValueError, RuntimeError, AssertionError) as exc:
            logger.info(f"fast captured step not built ({type(exc).__name__}: {exc})")
            return None
        self = cls()
        self.step, self.dev, self.states, self.mine = step, dev, states, mine
        self.feats, self.need, self.pack, self.n, self.padded = list(feats), tuple(need), pack, n, padded
        self.dec_params = list(model.decoder.parameters())
        self.hyper = (lr, (b1, b2), eps)
        self.bind(trainer, states)
        # kept with the model: the SLAM loop builds a new trainer (new optimizers) for every Mapper.mapping call, a few
        # iterations each -- a later trainer over the same grids adopts the plan (adopt()) instead of paying for a
        # new step, a new capture and a new plan every time
        plans = model.__dict__.setdefault('_fast_plans', [])
        plans.append(self)
        del plans[:-8]
        return self

    def bind(self, trainer, states):
        import weakref
        opt = trainer.optimizer
        self.states = states
        self.opt_ref = weakref.ref(opt)
        self.other_params = [p for g in opt.param_groups for p in g['params'] if not any(p is f for f in self.mine)]
        self.sig = self.signature(trainer)

    @classmethod
    def adopt(cls, trainer, n, padded):
        """A plan built by an earlier trainer of the same model that fits this trainer's (fresh) optimizer: same grids,
        same levels to train, same loss and Adam hyper-parameters.  The optimizer's state for those levels becomes the
        plan's buffers, zeroed -- what a new torch.optim.Adam starts from."""
        from miso_amd.grid_opt.loss import MisoLossMappingBase
        opt, lf, model = trainer.optimizer, trainer.loss_func, trainer.model
        plans = model.__dict__.get('_fast_plans')
        if not plans or type(opt) is not DenseAdam or opt._optimizer_step_pre_hooks or opt._optimizer_step_post_hooks:
            return None
        if not (isinstance(lf, MisoLossMappingBase) and type(lf).compute is MisoLossMappingBase.compute):
            return None
        for plan in reversed(plans):
            if plan.n != n or plan.padded != padded:
                continue
            old = plan.opt_ref()
            feats = plan.feats
            opt_params = {id(p) for group in opt.param_groups for p in group['params']}
            need = tuple(id(f) in opt_params and f.requires_grad for f in feats)
            if need != plan.need:
                continue
            groups = [g for g in opt.param_groups if any(any(p is f for f in plan.mine) for p in g['params'])]
            if {(g['lr'], tuple(g['betas']), g['eps']) for g in groups} != {plan.hyper}:
                continue
            states = [opt.state[f] for f in plan.mine]
            fresh = all(not st for st in states)
            if not fresh and old is not opt:
                continue                              # an optimizer with a history of its own: not ours to replace
            bufs = [a for a in plan.step.adam_state if a is not None]
            if not fresh and any(st.get('exp_avg') is not b[0] or st.get('exp_avg_sq') is not b[1] or st.get('active') is not b[2]
                                 for st, b in zip(states, bufs)):
                continue                              # its state no longer lives in the plan's buffers (a loaded state dict)
            if fresh and old is not None and old is not opt:
                # the optimizer that used the plan last is still alive (a trainer kept around, or one whose collection
                # is pending): it keeps its history in tensors of its own, the plan's buffers go to the new owner
                for st in plan.states:
                    for key in ('exp_avg', 'exp_avg_sq', 'active'):
                        if key in st:
                            st[key] = st[key].clone()
            if fresh:
                for st, (m, v, act) in zip(states, bufs):
                    m.zero_(); v.zero_(); act.zero_()
                    st.update(step=0, exp_avg=m, exp_avg_sq=v, active=act)
                plan.dev.set_count(0)
            saved = (plan.states, plan.opt_ref, plan.other_params, plan.sig)
            plan.bind(trainer, states)
            # everything else the capture baked in must still hold (loss scalars, flags, decoder weights ...): compare the
            # new signature with the old one field by field except for the optimizer's identity and state addresses
            if plan.sig[1:16] != saved[3][1:16]:
                plan.states, plan.opt_ref, plan.other_params, plan.sig = saved
                continue
            # the plan's step ADDS to the levels in _adam_clears and relies on its own Adam launch having left them
            # zero.  The buffers are shared with the checked-path steps of the trainers in between (share_grads), and a
            # binned checked step (optimizer.step(clear_grads=False)) leaves them non-zero: clear them here, once per
            # adoption (64 MB at the ScanNet shape, ~10 us), so the first replay does not add onto a stale gradient.
            for l, g in enumerate(plan.step.grads):
                if g is not None and (plan.step._adam_clears >> l) & 1:
                    g.zero_()
                    if plan.step.touched[l] is not None:
                        plan.step.touched[l].zero_()
            return plan
        return None

    def signature(self, trainer):
        """Everything the capture baked in, cheap to read: compared before every replay."""
        opt, lf, model = trainer.optimizer, trainer.loss_func, trainer.model
        return (id(opt), id(lf), id(model), lf.loss_type, float(lf.weight_sdf), float(lf.weight_fs), lf.trunc_dist,
                lf.weight_eik > 0, bool(lf.use_stability), lf.weight_clip > 0,
                tuple(bool(v) for v in model.ignore_level_),
                tuple((f.data_ptr(), f.requires_grad) for f in self.feats),
                tuple(p.requires_grad for p in self.dec_params),
                tuple(p.requires_grad for p in model.params_for_poses()),
                tuple((w.data_ptr(), w._version) for w in self.pack.weights),
                tuple((g['lr'], tuple(g['betas']), g['eps'], len(g['params'])) for g in opt.param_groups),
                tuple((st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), st['active'].data_ptr())
                      for st in self.states),
                len(opt._optimizer_step_pre_hooks), len(opt._optimizer_step_post_hooks))

    def run(self, trainer, model_input, gt, sanitize=False):
        from miso_amd import ops
        opt, model, step = trainer.optimizer, trainer.model, self.step
        coords_frame = model_input['coords_frame'][0]
        live = model_input.get('live_rows')
        if (coords_frame.shape[0] != self.n or not coords_frame.is_cuda or (live is not None) != self.padded
                or trainer.loss_func.__class__.compute is not _mapping_base_compute()
                or self.signature(trainer) != self.sig):
            return None
        # NaN guards of earlier steps that have arrived (a skipped step: the host counts go back, the device count never
        # moved); nothing here waits for the GPU
        self.dev.count -= opt.resolve_guard(block=False)
        count = self.states[0]['step']
        if any(st['step'] != count for st in self.states):
            return None
        if count != self.dev.count:
            self.dev.set_count(count)                         # a skipped step, a loaded state
        try:
            R_all, t_all = model.updated_kf_poses_all()
            with torch.no_grad():
                ops.mapping_batch(R_all, t_all.reshape(-1, 3), model.kf_key_index_table('KF'),
                                  model_input['sample_frame_ids'][0], coords_frame, gt['sdf'][0], gt['sdf_valid'][0],
                                  gt['sdf_signs'][0], model_input['weights'][0], step.x, step.aux, sanitize=sanitize)
        except (ValueError, AssertionError):
            return None                                       # a batch layout the launch does not take
        if live is not None:
            step.live_rows.copy_(live.reshape(1))
        step.run()
        # ---- what optimizer.step() does on the host ----------------------------------------------------------------
        for p in self.other_params:
            p.grad = None                                     # cf. the checked path: nothing stale may survive
        for f, g, nd in zip(self.feats, step.grads, self.need):
            if nd:
                if f.grad is not g:
                    f.grad = g
                torch.autograd.graph.increment_version(f)     # written through raw pointers
        for st in self.states:
            st['step'] = count + 1
        self.dev.count = count + 1
        if hasattr(opt, '_step_count'):
            opt._step_count += 1
        if step._use_graph:
            total = step.total.clone()                        # a replay writes the address the capture baked in
        else:
            # stream launches take the pointer per launch: hand this step's scalar out and give the next step a new one --
            # no copy kernel on the launch stream (4 us per step)
            total, step.total = step.total, torch.empty_like(step.total)
        # guards resolved while making room: those steps were skipped on the device, whose counter never moved
        self.dev.count -= opt.note_guarded_step(total, self.states, host=None if _GUARD_COPY else step.host_total)
        return total


_GUARD_COPY = os.environ.get('MISO_GUARD_COPY') is not None      # dev: the guard through a copy on the stream


def _mapping_base_compute():
    from miso_amd.grid_opt.loss import MisoLossMappingBase
    return MisoLossMappingBase.compute

class Trainer(object):
    def __init__(self, cfg, model, loss_func, train_dataloader, val_dataloader=None, device='cuda:0',
                 dtype=torch.float32):
        self.cfg = cfg
        self.verbose = cfg['verbose']
        self.model = model
        self.loss_func = loss_func
        self.use_cuda = torch.cuda.is_available()
        self.device = device
        self.train_dataloader = train_dataloader
        self.val_dataloader = val_dataloader
        self.model.to(self.device)
        self.set_optimizer()
        self.set_logging()

    # ---- setup -----------------------------------------------------------------------------
    def _load_pretrained(self):
        if self.cfg.get('pretrained_model') is not None:
            ckpt = torch.load(self.cfg['pretrained_model'])
            self.model.load_state_dict(ckpt['model_state_dict'])

    def set_optimizer(self):
        self._load_pretrained()
        self.optimizer = _make_optimizer(self.cfg['optimizer'], self.model.parameters(), self.cfg['learning_rate'])

    def set_external_optimizer(self, optimizer):
        self.optimizer = optimizer

    def set_logging(self):
        self.eval_metric = self.cfg.get('eval_metric')
        self.eval_best_loss = None
        self.eval_every = self.cfg['eval_every']
        self.ckpt_every = self.cfg['ckpt_every']
        self.log_dir = self.cfg['log_dir']
        self.ckpt_dir = os.path.join(self.log_dir, 'ckpt')
        self.tb_dir = os.path.join(self.log_dir, 'tensorboard')
        for d in (self.log_dir, self.ckpt_dir, self.tb_dir):
            cond_mkdir(d)
        self.train_dict = {'epochs': [], 'elapsed_time': [], 'epoch_time': [], 'total_loss': []}
        self.val_dict = {'epochs': [], 'total_loss': []}
        self.custom_eval_dict = {'epochs': []}
        self.custom_eval_funcs = dict()
        self.writer = SummaryWriter(self.tb_dir) if (SummaryWriter is not None and self.eval_every > 0) else None
        self.timer = PerfTimer(activate=True)

    def get_last_epoch(self):
        return self.train_dict['epochs'][-1] if self.train_dict['epochs'] else 0

    # ---- loop ---------------------------------------------------------------------------------
    def pre_epoch(self, epoch):
        if self.eval_every > 0 and epoch % self.eval_every == 0:
            self.run_eval(epoch)

    def post_epoch(self, epoch):
        if self.ckpt_every > 0 and epoch % self.ckpt_every == 0:
            self.save_model(epoch, f"ckpt_{epoch}")

    def train(self):
        self.total_steps = 0
        self.train_start_time = time.process_time()
        self.total_epoch_time = 0
        epoch = 0
        while epoch < self.cfg['epochs']:
            self.pre_epoch(epoch)
            self.train_epoch(epoch)
            self.post_epoch(epoch)
            epoch += 1
        if self.eval_every > 0:
            self.run_eval(epoch)
        if self.ckpt_every > 0:
            self.save_model(epoch, "final")

    # ---- captured mapping step -----------------------------------------------------------------
    def _captured_mapping_step(self, model_input, gt):
        """The common mapping configuration -- MisoLossMapping with only its sdf / free-space terms, a
        GridNet with a frozen MLP decoder, keyframe poses not optimised, dense Adam over feature
        grids -- as ONE graph replay (miso_amd.step.MappingStep: sort, forward + loss, backward, pull)
        followed by the usual optimizer.step().  Same arithmetic as the op-by-op path below; what
        goes away is ~60 launches and the autograd bookkeeping per iteration, which cost several
        times the 0.2 ms the GPU needs.  Returns the total loss, or None if the configuration
        does not qualify (the op-by-op path then runs)."""
        from miso_amd.grid_opt.loss import MisoLossMappingBase
        from miso_amd.step import MappingStep
        lf, model = self.loss_func, self.model
        fast = self.__dict__.get('_fast_plan')
        if fast is not None:
            total = fast.run(self, model_input, gt)
            if total is not None:
                return total
            self._fast_plan = None            # something changed (e.g. the coordinate schedule moved to another optimizer)
        if (self.cfg.get('fast_captured_step', True) and '_fast_plans' in model.__dict__
                and self.__dict__.get('_adopt_failed') != id(self.optimizer)):
            # a plan an earlier trainer of this model left behind (Mapper.mapping builds a trainer per call)
            cf = model_input['coords_frame'][0]
            fast = _FastMappingPlan.adopt(self, cf.shape[0], model_input.get('live_rows') is not None)
            if fast is not None:
                total = fast.run(self, model_input, gt)
                if total is not None:
                    self._fast_plan = fast
                    return total
            self._adopt_failed = id(self.optimizer)     # until the optimizer changes
        if not (isinstance(lf, MisoLossMappingBase) and type(lf).compute is MisoLossMappingBase.compute):
            return None
        if lf.loss_type not in ('L1', 'L2') or lf.weight_eik > 0 or lf.use_stability or lf.weight_clip > 0:
            return None
        if not isinstance(self.optimizer, DenseAdam) or not hasattr(model, '_fused_decoder'):
            return None
        coords_frame = model_input['coords_frame'][0]
        if not coords_frame.is_cuda or coords_frame.shape[0] == 0:
            return None
        pack = model._fused_decoder()
        if pack is None or any(p.requires_grad for p in model.decoder.parameters()):
            return None
        if any(p.requires_grad for p in model.params_for_poses()):
            return None
        feats = [g.feature for g in model.features]
        opt_params = {id(p) for group in self.optimizer.param_groups for p in group['params']}
        need = tuple(id(f) in opt_params and f.requires_grad for f in feats)
        if not any(need):
            return None
        # (a level that requires grad but is not in the active optimizer -- the coordinate schedule trains one level at
        # a time -- gets no gradient here.  Autograd would accumulate one that nothing consumes: the level's own
        # optimizer starts with zero_grad() when its turn comes, trainer.py:206.  Parameters are the same either way.)
        n = coords_frame.shape[0]
        live = model_input.get('live_rows')          # padded batch (datasets with padded=True): count on the device
        ignore = tuple(bool(v) for v in model.ignore_level_)
        key = (n, live is not None, need, ignore, lf.loss_type, float(lf.weight_sdf), float(lf.weight_fs), lf.trunc_dist,
               tuple(f.data_ptr() for f in feats))
        cache = self.__dict__.setdefault('_mapping_steps', {})
        step = cache.get(key)
        if step is None:
            # a new key.  Datasets with a data-dependent row count (PosedSdfRgbd(padded=False): depth holes) change
            # n almost every batch: building a step and capturing a graph that is never replayed would cost more
            # than the op-by-op path.  So the step of a new key runs its launches eagerly and shares the gradient
            # buffers of the previous one; the graph is captured only when the same key comes back next time.
            prev = next(iter(cache.values()), None)
            cache.clear()            # one live step: batch sizes rarely alternate
            meta = model.features[0].grid_meta(model.ignore_level_)
            step = MappingStep([f.data for f in feats], meta, pack, n, lf.loss_type, float(lf.weight_sdf),
                               float(lf.weight_fs) if lf.weight_fs > 0 else 0.0,
                               0.0 if lf.trunc_dist is None else float(lf.trunc_dist), need_levels=need,
                               keep_sdf=False, padded=live is not None, grads_cleared_by_optimizer=True,
                               use_graph=False, crowded=bool(self.cfg.get('crowded_batches', False)),
                               share_grads=None if prev is None or prev.need_levels != list(need) else prev.grads)
            cache[key] = step
        elif not step._use_graph and not step.__dict__.get('_seen_again'):
            step._seen_again = True
            # same batch shape twice in a row: from now on one graph replay per step (small batches; large ones stay
            # on the stream, MappingStep.STREAM_MIN_POINTS)
            step._use_graph = step.n < step.STREAM_MIN_POINTS
            if self.cfg.get('fast_captured_step', True):
                # ... and from the step after this one, with the optimizer inside the replay (_FastMappingPlan)
                self._fast_plan_due = key
        with torch.no_grad():
            frame_ids = model_input['sample_frame_ids'][0, :, 0]
            coords_world = lf.world_coords(model, coords_frame, frame_ids)
            step.set_batch(coords_world, gt['sdf'][0], gt['sdf_valid'][0], gt['sdf_signs'][0],
                           model_input['weights'][0], live_rows=live)
        step.run()
        # the reference's optimizer.zero_grad(set_to_none=True) (trainer.py:206): Adam steps every parameter whose
        # .grad is not None whatever its requires_grad, so a stale gradient on anything this step does not write
        # (keyframe pose corrections left over from an adam tracking window, a locked level) must not survive
        mine = {id(f) for f, nd in zip(feats, need) if nd}
        for group in self.optimizer.param_groups:
            for p in group['params']:
                if id(p) not in mine:
                    p.grad = None
        for f, g, nd in zip(feats, step.grads, need):
            f.grad = g if nd else None
        total = step.loss.sum()
        # small batches scatter into the step's persistent gradient buffers: the optimizer clears what it
        # consumed in the same pass (a memset of a 0.5 GB level costs as much as the rest of the step)
        clear = step.sorted is None
        # NaN guard (reference :213-219) on the device: the optimizer's kernels leave everything alone if the
        # loss is NaN and the host hears about it one step later -- no read-back between backward and step
        # the step's scatter kernels flagged the 64-float chunks they wrote: Adam reads the flags, not the gradient
        self.optimizer.step(clear_grads=clear, guard=total,
                            touched={id(f): t for f, t, nd in zip(feats, step.touched, need) if nd})
        if self.__dict__.pop('_fast_plan_due', None) == key:
            self._fast_plan = _FastMappingPlan.build(self, step, feats, need, pack, n, live is not None)
        return total

    def train_step(self, model_input, gt, _raw=False):
        """zero_grad -> loss dict -> sum of means -> NaN guard -> backward -> step.
        Returns the total loss (device scalar).  _raw (train_epoch): the batch has been moved to the device but not
        been through nan_to_num yet -- the one-replay plan folds that into its first launch, every other path
        sanitises here."""
        if self.cfg.get('captured_step', True):
            if _raw:
                fast = self.__dict__.get('_fast_plan')
                total = fast.run(self, model_input, gt, sanitize=True) if fast is not None else None
                if total is not None:
                    return total
                from .utils.utils import sanitize_tensor_dict
                model_input, gt = sanitize_tensor_dict(model_input), sanitize_tensor_dict(gt)
                if fast is not None:
                    self._fast_plan = None
            total = self._captured_mapping_step(model_input, gt)
            if total is not None:
                return total
        elif _raw:
            from .utils.utils import sanitize_tensor_dict
            model_input, gt = sanitize_tensor_dict(model_input), sanitize_tensor_dict(gt)
        self.optimizer.zero_grad()
        loss_dict = self.loss_func.compute(self.model, model_input, gt)
        total = 0.
        for value in loss_dict.values():
            total = total + value.mean()
        if not torch.isnan(total):
            total.backward(retain_graph=False)
            self.optimizer.step()
        else:
            logger.warning("Loss is nan! Skip backward step.")
        return total

    # ---- device time of the training steps (reference: PerfTimer.check() after every step, trainer.py:131-140) --------
    # The reference synchronises the device after every step to read its timer, which puts the host's work for the next
    # step behind the GPU's for this one.  Here every step is bracketed by two events and the elapsed times are added
    # up when somebody reads total_epoch_time (eval / checkpoint time): same number, no wait inside the loop.
    @property
    def total_epoch_time(self):
        self._settle_step_times(block=True)
        return self.__dict__.get('_epoch_time', 0)

    @total_epoch_time.setter
    def total_epoch_time(self, value):
        self.__dict__['_step_events'] = []
        self.__dict__['_epoch_time'] = value

    def _settle_step_times(self, block):
        events = self.__dict__.get('_step_events')
        while events:
            start, end = events[0]
            if isinstance(start, float):                       # no device: wall clock
                dt = end - start
            else:
                if block:
                    end.synchronize()
                elif not end.query():
                    break
                dt = start.elapsed_time(end) / 1e3
            events.pop(0)
            self.__dict__['_epoch_time'] = self.__dict__.get('_epoch_time', 0) + dt

    @staticmethod
    def _batches(loader):
        from .utils.utils import iter_batches
        return iter_batches(loader)

    def train_epoch(self, epoch):
        # model.train() walks every submodule (65 us on a 100-keyframe GridNet) and an epoch here is ONE step: skip the
        # walk when the model and its direct children are in training mode already
        if not (self.model.training and all(m.training for m in self.model.children())):
            self.model.train()
        on_gpu = torch.cuda.is_available()
        events = self.__dict__.setdefault('_step_events', [])
        for step, (model_input, gt) in enumerate(self._batches(self.train_dataloader)):
            if on_gpu:
                start = torch.cuda.Event(enable_timing=True)
                start.record()
            else:
                start = time.perf_counter()
            if type(self).train_step is Trainer.train_step:
                model_input, gt = prepare_batch(model_input, gt, self.device, sanitize=False)
                total = self.train_step(model_input, gt, _raw=True)
            else:       # a subclass with its own train_step(model_input, gt): the reference's call, sanitised batch
                model_input, gt = prepare_batch(model_input, gt, self.device)
                total = self.train_step(model_input, gt)
            self.total_steps += 1
            if self.verbose and step % 10 == 0:
                logger.info(f"Train epoch {epoch} step {step} | train_loss={float(total.detach()):.2e}.")
            if on_gpu:
                end = torch.cuda.Event(enable_timing=True)
                end.record()
            else:
                end = time.perf_counter()
            events.append((start, end))
            if len(events) > 64:
                self._settle_step_times(block=False)

    def relative_param_change(self, epoch, params_list):
        self.params_curr = [p.clone().detach() for p in params_list]
        if self.params_prev is None:
            self.params_prev = self.params_curr
            return np.inf
        num = sum(torch.sum((c - p) ** 2) for c, p in zip(self.params_curr, self.params_prev))
        den = sum(torch.sum(p ** 2) for p in self.params_prev)
        self.params_prev = self.params_curr
        return torch.sqrt(num / den)

    # ---- eval / checkpoints ------------------------------------------------------------------------
    def register_eval_func(self, name, func):
        self.custom_eval_funcs[name] = func
        self.custom_eval_dict[name] = []

    def run_eval(self, epoch):
        self.eval(epoch, 'train')
        self.eval(epoch, 'val')
        self.custom_eval_dict['epochs'].append(epoch)
        for name, func in self.custom_eval_funcs.items():
            self.custom_eval_dict[name].append(
                func(epoch, self.cfg, self.model, self.loss_func, self.train_dataloader, self.val_dataloader))

    def eval(self, epoch, mode='train'):
        self.model.eval()
        if mode == 'train':
            loader, target = self.train_dataloader, self.train_dict
        elif mode == 'val':
            loader, target = self.val_dataloader, self.val_dict
        else:
            raise ValueError(f"Invalid eval mode: {mode}!")
        if loader is None:
            return
        sums = {}
        for model_input, gt in loader:
            model_input, gt = self.prepare_batch(model_input, gt)
            for name, value in self.loss_func.compute(self.model, model_input, gt).items():
                sums.setdefault(name, []).append(value.mean().item())
        target['epochs'].append(epoch)
        total = 0.0
        for name, vals in sums.items():
            avg = float(np.mean(np.asarray(vals)))
            target.setdefault(name, []).append(avg)
            total += avg
            if self.writer is not None:
                self.writer.add_scalar(f"{mode}/{name}", avg, epoch)
        target['total_loss'].append(total)
        if mode == 'train':
            target['elapsed_time'].append(time.process_time() - self.train_start_time)
            target['epoch_time'].append(self.total_epoch_time)
        if mode == 'val' and self.eval_metric is not None:
            cur = target[self.eval_metric][-1]
            if self.eval_best_loss is None or self.eval_best_loss > cur:
                self.eval_best_loss = cur
                self.save_model(epoch, 'best_model')

    def save_model(self, epoch, ckpt_name):
        """Same checkpoint dict as the reference (trainer.py:319-332)."""
        torch.save({'epoch': epoch, 'model_state_dict': self.model.state_dict(),
                    'optimizer_state_dict': self.optimizer.state_dict(), 'train_dict': self.train_dict,
                    'val_dict': self.val_dict}, os.path.join(self.ckpt_dir, f"{ckpt_name}.pt"))
        if callable(getattr(self.model, 'save', None)):
            self.model.save(self.ckpt_dir, ckpt_name)

    def prepare_batch(self, model_input, gt):
        model_input = {k: v.to(self.device) for k, v in model_input.items()}
        if 'coords' in model_input:
            model_input['coords'].requires_grad_(True)
        gt = {k: v.to(self.device) for k, v in gt.items()}
        return model_input, gt


class GridTrainer(Trainer):
    """Per-level Adam optimisers switched every ``max_epochs_in_level`` epochs (or on
    convergence), then an optional joint optimiser (reference trainer.py:370-480)."""

    def reset_convergence_check(self):
        self.params_prev = None
        self.params_curr = None
        self.relchange = np.inf
        self.epochs_in_level = 0

    def set_optimizer(self):
        self.relchange_tol = self.cfg['relchange_tol']
        self.max_epochs_in_level = self.cfg['max_epochs_in_level']
        self.grid_training_mode = self.cfg['grid_training_mode']
        self._load_pretrained()
        name, lr = self.cfg['optimizer'], self.cfg['learning_rate']
        if name not in ('adam', 'sgd'):
            raise ValueError(f"Invalid optimizer: {name}.")
        self.level_optimizers = []
        if self.grid_training_mode != 'joint':
            for level in range(self.model.num_levels):
                self.level_optimizers.append(_make_optimizer(name, self.model.params_at_level(level), lr))
        self.joint_optimizer = _make_optimizer(name, self.model.parameters(), lr)
        self.reset_convergence_check()
        if self.grid_training_mode in ('coordinate', 'coordinate+joint'):
            self.active_level = 0
            self.optimizer = self.level_optimizers[0]
        elif self.grid_training_mode == 'joint':
            self.active_level = self.model.num_levels
            self.optimizer = self.joint_optimizer
        else:
            raise ValueError(f"Invalid grid training mode: {self.grid_training_mode}")

    def pre_epoch(self, epoch):
        super().pre_epoch(epoch)
        done = self.relchange < self.relchange_tol or self.epochs_in_level >= self.max_epochs_in_level
        if done and self.active_level < self.model.num_levels:
            self.train_dict[f'level{self.active_level}_last_epoch'] = epoch
            self.active_level += 1
            if self.active_level >= self.model.num_levels:
                if self.grid_training_mode == 'coordinate+joint':
                    self.optimizer = self.joint_optimizer
            else:
                self.optimizer = self.level_optimizers[self.active_level]
            self.reset_convergence_check()
        self.epochs_in_level += 1

    def eval(self, epoch, mode='train'):
        super().eval(epoch, mode)
        if mode == 'train':
            self.relchange = self.relative_param_change(epoch, self.model.params_at_level(self.active_level))
            self.train_dict.setdefault('relchange', []).append(self.relchange)
