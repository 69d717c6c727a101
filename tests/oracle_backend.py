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


class AlignPlan:
    """TEST-ONLY torch restatement of ops.AlignPlan (miso_align_iteration_a / _b) so that the host side of
    base.fused_alignment_loop and the sharded loop of miso_amd.dist run without a GPU.  Follows
    generic_align_multiple_submaps (grid_opt/align/base.py:89-163) step by step with autograd."""
    cpu_ok = True

    def __init__(self, R0, t0, pairs, *, loss_type="L2", align_weight=3000.0, overlap_thresh=1e-2, lr=1e-2,
                 betas=(0.9, 0.999), eps=1e-8, reg_weight=0.0, reg_thresh_rad=1.0, reg_thresh_m=1.0,
                 rel_change_thresh=0.0, ring_iters=0, save_poses=False):
        self.S, self.P = R0.shape[0], len(pairs)
        self.R0, self.t0, self.pairs = R0.detach().clone(), t0.detach().clone(), pairs
        self.k = dict(loss_type=loss_type, w=float(align_weight), thr=float(overlap_thresh), lr=float(lr), b1=betas[0],
                      b2=betas[1], eps=eps, reg=float(reg_weight), rad=float(reg_thresh_rad), m=float(reg_thresh_m),
                      rel=float(rel_change_thresh))
        self.params = torch.zeros(self.S, 6)
        self.flat_reduce = torch.zeros(7 * self.S + 1)      # flat + per-submap counts of pairs that passed the gate
        self.flat = self.flat_reduce[:6 * self.S + 1]
        self.steps = [0] * self.S
        self.pair_losses = torch.zeros(self.P)
        self.m, self.v = torch.zeros(self.S, 6), torch.zeros(self.S, 6)
        self.ring_iters, self.save_poses = int(ring_iters), bool(save_poses)
        self._ring = torch.zeros(self.ring_iters, 2 + (16 * self.S if save_poses else 0))
        self._c = dict(steps=0, stopped=False, iterations=0, skipped=0)

    def iteration_a(self):
        if self._c['stopped']:
            return
        k = self.k
        dr = self.params[:, :3].clone().requires_grad_(True)
        dt = self.params[:, 3:].clone().requires_grad_(True)
        Rm = self.R0 @ R.so3_exp_map(dr)
        tm = self.t0 + dt.reshape(-1, 3, 1)
        it = self._c['iterations']
        if self.save_poses and it < self.ring_iters:
            T = torch.zeros(self.S, 4, 4)
            T[:, :3, :3], T[:, :3, 3:], T[:, 3, 3] = Rm.detach(), tm.detach(), 1.0
            self._ring[it, 2:] = T.reshape(-1)
        total = torch.zeros(())
        had = torch.zeros(self.S)
        for i, pr in enumerate(self.pairs):
            a, b = pr["src"], pr["dst"]
            bound = _bound(pr["meta_dst"])
            gate = 1.0
            if pr.get("gate_pts") is not None:
                with torch.no_grad():
                    q = R.transfrom_points_from(R.transform_points_to(pr["gate_pts"], Rm[a], tm[a]), Rm[b], tm[b])
                    gate = float((torch.count_nonzero(R.coords_in_bound(q, bound)) / q.shape[0]) > k['thr'])
            q = R.transfrom_points_from(R.transform_points_to(pr["coords"], Rm[a], tm[a]), Rm[b], tm[b])
            inb = R.coords_in_bound(q, bound)
            w = inb.to(q.dtype)
            q = torch.where(inb, q, torch.zeros_like(q))      # NaN / far-away rows carry no weight: keep the gather finite
            F = sum(int(f.shape[1]) for f in pr["feats_dst"])
            diff = (pr["feats_src"][:, :F] - encode(q, pr["feats_dst"], pr["meta_dst"])) * w
            n_valid = w.sum().clamp(min=1.0)
            val = diff.pow(2).sum() / (n_valid * F) if k['loss_type'] == "L2" \
                else torch.linalg.vector_norm(diff, dim=1).sum() / n_valid
            val = torch.nan_to_num(val) * k['w'] * gate
            had[a] += gate
            had[b] += gate
            self.pair_losses[i] = val.detach()
            total = total + val
        self.flat_reduce.zero_()
        self.flat_reduce[6 * self.S + 1:] = had
        if total.requires_grad:
            gr, gt = torch.autograd.grad(total, (dr, dt), allow_unused=True)
            g = torch.cat((torch.zeros_like(dr) if gr is None else gr, torch.zeros_like(dt) if gt is None else gt), 1)
            self.flat[:6 * self.S] = g.reshape(-1)
        self.flat[6 * self.S] = total.detach()

    def iteration_b(self):
        c, k = self._c, self.k
        if c['stopped']:
            return
        p = self.params.clone().requires_grad_(True)
        total = self.flat[6 * self.S].clone()
        g = self.flat[:6 * self.S].reshape(self.S, 6).clone()
        if k['reg'] > 0:
            reg = sum(k['reg'] * torch.relu(torch.linalg.norm(p[s, :3]) - k['rad'])
                      + k['reg'] * torch.relu(torch.linalg.norm(p[s, 3:]) - k['m']) for s in range(self.S))
            total = total + reg.detach()
            if reg.requires_grad:
                g = g + torch.nan_to_num(torch.autograd.grad(reg, p)[0])
        old = self.params[1:].clone()
        if not bool(torch.isnan(total)):
            c['steps'] += 1
            # torch.optim.Adam per pose tensor: a submap without a gradient (none of its pairs passed the overlap gate,
            # no regulariser) is skipped -- value, moments, its own step count
            for s in range(1, self.S):
                if not (k['reg'] > 0 or float(self.flat_reduce[6 * self.S + 1 + s]) > 0):
                    continue
                self.steps[s] += 1
                t = self.steps[s]
                m, v, gg = self.m[s], self.v[s], g[s]
                m.lerp_(gg, 1 - k['b1'])
                v.mul_(k['b2']).addcmul_(gg, gg, value=1 - k['b2'])
                denom = (v.sqrt() / (1 - k['b2'] ** t) ** 0.5).add_(k['eps'])
                self.params[s] = self.params[s].addcdiv(m, denom, value=-(k['lr'] / (1 - k['b1'] ** t)))
        else:
            c['skipped'] += 1
        it = c['iterations']
        rel = float('inf') if it == 0 else float(torch.sqrt(((self.params[1:] - old) ** 2).sum() / (old ** 2).sum()))
        if it < self.ring_iters:
            self._ring[it, 0], self._ring[it, 1] = total, rel
        if rel < k['rel']:
            c['stopped'] = True
        c['iterations'] = it + 1

    def ctrl(self):
        return dict(self._c)

    def ring(self):
        return self._ring


def install(monkeypatch):
    from miso_amd import ops
    monkeypatch.setattr(ops, "encode", encode)
    monkeypatch.setattr(ops, "grid_sample_3d", grid_sample_3d)
    monkeypatch.setattr(ops, "mapping_loss", mapping_loss)
    monkeypatch.setattr(ops, "sdf_fused_supported", lambda *a, **k: False)
    monkeypatch.setattr(ops, "AlignPlan", AlignPlan)
