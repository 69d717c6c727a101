"""iSDF-style bound loss (reference: grid_opt/loss_isdf.py).  The eikonal / gradient /
smoothness variants differentiate a spatial gradient obtained with create_graph=True;
that is what drives the second-order encode kernel (miso_encode_bwd2)."""
import torch
import torch.nn.functional as F
from torch.autograd import grad

from .loss import (BaseLoss, compute_feature_regularization_loss, compute_feature_stability_loss,
                   compute_pose_trust_region_loss, transform_by_keyframe)
from .models.base_net import BaseNet
from .models.grid_atlas import GridAtlas
from .models.grid_net import GridNet


def gradient(inputs, outputs):
    """d outputs / d inputs with the graph kept (reference :367-377)."""
    from miso_amd import ops
    with ops.coordinate_gradient_only():          # (only d outputs / d inputs is asked for: no grid gradients on the way)
        return grad(outputs=outputs, inputs=inputs, grad_outputs=torch.ones_like(outputs), create_graph=True,
                    retain_graph=True, only_inputs=True)[0]


def full_sdf_loss(sdf, target_sdf, free_space_factor=5.0):
    """Reference :280-296: free-space branch max(relu(s-b), exp(-5 s)-1), truncation branch s-b."""
    free = torch.max(F.relu(sdf - target_sdf), torch.exp(-free_space_factor * sdf) - 1.)
    return free, sdf - target_sdf


def sdf_loss(sdf, bounds, t, loss_type="L1", p75=0.05):
    """Reference :299-333."""
    free, trunc = full_sdf_loss(sdf, bounds)
    free_space_ixs = bounds > t
    zero = torch.zeros_like(free)
    mat = torch.where(free_space_ixs, free, zero) + torch.where(free_space_ixs, zero, trunc)
    if loss_type == "L1":
        mat = torch.abs(mat)
    elif loss_type == "L2":
        mat = torch.square(mat)
    elif loss_type == "GM":
        raise ValueError("GM loss is deprecated.")
    else:
        raise ValueError("Must be L1 or L2")
    return mat, free_space_ixs


def tot_loss(sdf_loss_mat, grad_loss_mat, eik_loss_mat, free_space_ixs, bounds, eik_apply_dist,
             trunc_weight, grad_weight, eik_weight):
    """Reference :335-365.  ``losses`` holds device scalars (the reference calls .item()
    here, a host sync per iteration)."""
    sdf_loss_mat = torch.where(free_space_ixs, sdf_loss_mat, sdf_loss_mat * trunc_weight)
    losses = {"sdf_loss": sdf_loss_mat.mean().detach()}
    total = sdf_loss_mat
    if grad_loss_mat is not None:
        total = total + grad_weight * grad_loss_mat
        losses["grad_loss"] = grad_loss_mat.mean().detach()
    if eik_loss_mat is not None:
        eik = torch.where(bounds.squeeze(-1) < eik_apply_dist, torch.zeros_like(eik_loss_mat), eik_loss_mat)
        eik = eik * eik_weight
        total = total + eik
        losses["eikonal_loss"] = eik.mean().detach()
    tot = total.mean()
    losses["total_loss"] = tot
    return tot, total, losses


class iSDFLoss(BaseLoss):
    def __init__(self, model_name, trunc_weight, trunc_distance, noise_std=0, orien_loss=0, eik_apply_dist=0.1,
                 eik_weight=0, grad_weight=0, smooth_weight=0.0, smooth_std=0.05, loss_type="L1",
                 slam_mode=True, pose_reg_weight=0, pose_thresh_rad=1.0, pose_thresh_m=1.0, residual_p75=0.05):
        super().__init__()
        self.model_name = model_name
        self.trunc_weight = trunc_weight
        self.trunc_distance = trunc_distance
        self.noise_std = noise_std
        self.orien_loss = orien_loss
        self.eik_apply_dist = eik_apply_dist
        self.eik_weight = eik_weight
        self.grad_weight = grad_weight
        self.smooth_weight = smooth_weight
        self.smooth_std = smooth_std
        self.loss_type = loss_type
        self.cosSim = torch.nn.CosineSimilarity(dim=-1, eps=1e-6)
        self.slam_mode = slam_mode
        self.pose_reg_weight = pose_reg_weight
        self.pose_thresh_rad = pose_thresh_rad
        self.pose_thresh_m = pose_thresh_m
        self.residual_p75 = residual_p75

    def compute(self, model: BaseNet, model_input: dict, gt: dict):
        return self.compute_slam(model, model_input, gt) if self.slam_mode \
            else self.compute_default(model, model_input, gt)

    def _smooth(self, model, pc, sdf):
        pc_p = (pc + torch.randn_like(pc) * self.smooth_std).requires_grad_()
        sdf_p = model(pc_p[0], noise_std=self.noise_std)
        g_p, g = gradient(pc_p, sdf_p), gradient(pc, sdf)
        return self.smooth_weight * torch.mean((g_p[:, :, 0] - g[:, :, 0]).norm(2, dim=-1))

    def compute_slam(self, model, model_input: dict, gt: dict):
        """Reference :46-93."""
        pc_kf = model_input['coords_frame'][0]
        kf_idxs = model_input['sample_frame_ids'][0, :, 0]
        pc = transform_by_keyframe(pc_kf, kf_idxs, model.updated_kf_pose_in_world).unsqueeze(0)
        bounds = gt['sdf']
        sdf = model(pc[0], noise_std=self.noise_std)
        mat, free_ixs = sdf_loss(sdf, bounds, self.trunc_distance, loss_type=self.loss_type, p75=self.residual_p75)
        total, _, _ = tot_loss(mat, None, None, free_ixs, bounds, self.eik_apply_dist, self.trunc_weight,
                               self.grad_weight, self.eik_weight)
        loss_dict = {"sdf": total}
        if self.smooth_weight > 0:
            loss_dict['smooth'] = self._smooth(model, pc, sdf)
        if self.pose_reg_weight > 0:
            rot = torch.linalg.norm(model.rotation_corrections, dim=1)
            tran = torch.linalg.norm(model.translation_corrections.squeeze(2), dim=1)
            loss_dict['trust_region_R'] = self.pose_reg_weight * torch.sum(F.relu(rot - self.pose_thresh_rad))
            loss_dict['trust_region_t'] = self.pose_reg_weight * torch.sum(F.relu(tran - self.pose_thresh_m))
        return loss_dict

    def compute_default(self, model: BaseNet, model_input: dict, gt: dict):
        """Reference :96-152."""
        pc, norm_sample = model_input['coords'], model_input['normals']
        bounds, grad_vec = gt['sdf'], gt['grad_vec']
        need_grad = self.eik_weight != 0 or self.grad_weight != 0 or self.smooth_weight != 0
        if need_grad:
            pc.requires_grad_()
        sdf = model(pc[0], noise_std=self.noise_std)
        sdf_grad = gradient(pc, sdf) if need_grad else None
        mat, free_ixs = sdf_loss(sdf, bounds, self.trunc_distance, loss_type=self.loss_type, p75=self.residual_p75)
        eik_mat = torch.abs(sdf_grad.norm(2, dim=-1) - 1) if self.eik_weight != 0 else None
        grad_mat = None
        if self.grad_weight != 0:
            sdf_grad = sdf_grad.reshape(sdf_grad.shape[0], grad_vec.shape[1], -1, 3)
            surf = 1 - self.cosSim(sdf_grad[:, :, 0], norm_sample)
            nan_rows = torch.where(grad_vec[..., 0].isnan())
            grad_vec[nan_rows] = norm_sample[nan_rows[:2]]
            grad_mat = torch.cat((surf[:, :, None], 1 - self.cosSim(grad_vec, sdf_grad[:, :, 1:])), dim=2)
            grad_mat = grad_mat.reshape(grad_mat.shape[0], -1, 1)
            if self.orien_loss:
                grad_mat = (grad_mat > 1).float()
        total, _, _ = tot_loss(mat, grad_mat, eik_mat, free_ixs, bounds, self.eik_apply_dist,
                               self.trunc_weight, self.grad_weight, self.eik_weight)
        loss_dict = {"sdf": total}
        if self.smooth_weight > 0:
            loss_dict['smooth'] = self._smooth(model, pc, sdf)
        return loss_dict


class iSDFLossSubmap(BaseLoss):
    """Per-submap iSDF loss over a GridAtlas (reference :155-277)."""

    def __init__(self, model_name, trunc_weight, trunc_distance, noise_std, orien_loss, eik_apply_dist,
                 eik_weight=0, grad_weight=0, smooth_weight=0.1, smooth_std=0.05, loss_type="L1",
                 feat_reg_weight=0, slam_mode=False, pose_reg_weight=1e3, pose_thresh_rad=1.0,
                 pose_thresh_m=1.0):
        super().__init__()
        self.model_name = model_name
        self.trunc_weight = trunc_weight
        self.trunc_distance = trunc_distance
        self.noise_std = noise_std
        self.orien_loss = orien_loss
        self.eik_apply_dist = eik_apply_dist
        self.eik_weight = eik_weight
        self.grad_weight = grad_weight
        self.smooth_weight = smooth_weight
        self.smooth_std = smooth_std
        self.loss_type = loss_type
        self.cosSim = torch.nn.CosineSimilarity(dim=-1, eps=1e-6)
        self.feat_reg_weight = feat_reg_weight
        self.pose_reg_weight = pose_reg_weight
        self.pose_thresh_rad = pose_thresh_rad
        self.pose_thresh_m = pose_thresh_m
        self.slam_mode = slam_mode

    def compute_submap_loss(self, model, model_input: dict, gt: dict):
        pc = model_input['coords']
        bounds = gt['sdf']
        need_grad = self.eik_weight != 0 or self.grad_weight != 0
        if need_grad:
            pc.requires_grad_()
        sdf = model(pc[0], noise_std=self.noise_std)
        sdf_grad = gradient(pc, sdf) if need_grad else None
        mat, free_ixs = sdf_loss(sdf, bounds, self.trunc_distance, loss_type=self.loss_type)
        eik_mat = torch.abs(sdf_grad.norm(2, dim=-1) - 1) if self.eik_weight != 0 else None
        if self.grad_weight != 0 or self.smooth_weight > 0:
            raise NotImplementedError
        total, _, _ = tot_loss(mat, None, eik_mat, free_ixs, bounds, self.eik_apply_dist, self.trunc_weight,
                               self.grad_weight, self.eik_weight)
        loss_dict = {"sdf": total}
        if isinstance(model, GridNet):
            loss_dict.update(compute_feature_stability_loss(model, pc[0]))
            if self.feat_reg_weight > 0:
                loss_dict.update(compute_feature_regularization_loss(model, weight=self.feat_reg_weight))
            if self.pose_reg_weight > 0:
                loss_dict.update(compute_pose_trust_region_loss(model, thresh_rad=self.pose_thresh_rad,
                                                                thresh_m=self.pose_thresh_m,
                                                                weight=self.pose_reg_weight))
        return loss_dict

    def compute(self, model: GridAtlas, model_input: dict, gt: dict):
        losses = {}
        owner = model_input['submap_idxs'][0, :, 0]
        for s in range(model.num_submaps):
            rows = torch.nonzero(owner == s, as_tuple=False).squeeze(1)
            if rows.numel() == 0:
                continue
            sub_gt = {'sdf': gt['sdf'][:, rows, :], 'sdf_valid': gt['sdf_valid'][:, rows, :]}
            if self.slam_mode:
                coords_kf = model_input['coords_kf'][0, rows, :]
                kf_idxs = model_input['keyframe_idxs'][0, rows, 0]
                coords = transform_by_keyframe(coords_kf, kf_idxs,
                                               lambda k, _s=s: model.updated_kf_pose_in_submap(k, _s))
                sub_in = {'coords': coords.unsqueeze(0)}
            else:
                sub_in = {'coords': model_input['coords_submap'][:, rows, :]}
            for key, val in self.compute_submap_loss(model.get_submap(s), sub_in, sub_gt).items():
                losses[f'submap{s}_{key}'] = val
        return losses
