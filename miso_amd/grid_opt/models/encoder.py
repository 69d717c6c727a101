"""Learned initialisation of a GridNet (reference: grid_opt/models/encoder.py).

Level by level, coarse to fine: the SDF residuals of the current grid at the observed samples are pooled onto the
level's lattice (three scatter-averages: SDF error, free-space upper and lower violations), a small 3-D conv net
(FeaturePrediction) turns that volume into a feature correction for the level, and the next level sees the
residuals that remain.  The residual queries go through the fused multi-level encode (second-order capable, as the
reference's ``second_order_grid_sample=True``) and the frozen decoder.  Pretrained predictor weights are not
shipped with the reference; state-dict keys match, so upstream's ``feature_encoder_level_{l}.pt`` load as they are.
"""
import logging
from dataclasses import dataclass
from os.path import join

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from miso_amd import ops
import miso_amd.grid_opt.utils.utils as utils
import miso_amd.grid_opt.utils.utils_geometry as utils_geometry
from miso_amd.grid_opt.loss import BaseLoss
from .grid_net import GridNet
from .modules import FeaturePrediction

logger = logging.getLogger(__name__)


@dataclass
class EncoderObservation:
    """Raw SDF observations the encoder consumes (reference :16-22)."""
    coords_world: Tensor   # (N, 3)
    gt_sdf: Tensor         # (N, 1)
    gt_sdf_sign: Tensor    # (N, 1)  1 = free space
    gt_sdf_valid: Tensor   # (N, 1)  1 = gt_sdf usable


class Encoder(torch.nn.Module):
    def __init__(self, cfg, pretrained_dir=None):
        super().__init__()
        self.num_levels = cfg['model']['grid']['n_levels']
        self.fdim = cfg['model']['grid']['feature_dim']
        self.rdim = 3                 # pooled residual channels: sdf, free-space upper, free-space lower
        self.trunc_dist = 0.15        # fixed at pre-training time upstream (:38)
        self.device = cfg['device']
        assert self.num_levels == 2
        self.feature_encoders = torch.nn.ModuleList(
            self.init_level_encoder(join(pretrained_dir, f"feature_encoder_level_{l}.pt") if pretrained_dir else None)
            for l in range(self.num_levels))
        self.grid_nets = torch.nn.ModuleDict()
        self.grid_corrections = torch.nn.ParameterDict()
        self.intermediate_results = {}

    def init_level_encoder(self, pretrained_path=None):
        enc = FeaturePrediction(d=3, fdim=self.fdim, rdim=self.rdim, feature_processor=False, residual_processor=True,
                                device=self.device)
        if pretrained_path is not None:
            enc.load_state_dict(torch.load(pretrained_path))
        for p in enc.parameters():
            p.requires_grad = False
        return enc

    # ---- bookkeeping (reference :65-119) ----------------------------------------------------
    def lock_all_params(self):
        for p in self.feature_encoders.parameters():
            p.requires_grad = False
        for model in self.grid_nets.values():
            for level in range(model.num_levels):
                model.lock_level(level)
        for corr in self.grid_corrections.values():
            corr.requires_grad = False

    def unlock_encoder_at_level(self, level):
        for p in self.feature_encoders[level].parameters():
            p.requires_grad = True

    def grid_key(self, model_id):
        return f"gridnet{model_id}"

    def correction_key(self, model_id, level):
        return f"gridnet{model_id}_correction_level{level}"

    def get_grid_net(self, model_id) -> GridNet:
        return self.grid_nets[self.grid_key(model_id)]

    def get_grid_correction(self, model_id, level) -> torch.nn.Parameter:
        return self.grid_corrections[self.correction_key(model_id, level)]

    def register_grid_model(self, model: GridNet):
        model_id = len(self.grid_nets)
        self.grid_nets[self.grid_key(model_id)] = model
        for level in range(model.num_levels):
            self.grid_corrections[self.correction_key(model_id, level)] = torch.nn.Parameter(
                torch.zeros_like(model.features[level].feature))
        return model_id

    def print_trainable_params(self):
        print("=== Summary of trainable params === ")
        for name, p in self.named_parameters():
            if p.requires_grad:
                print(f"{name}: {p.shape}")
        print("=== END Summary of trainable params ===")

    # ---- prediction -------------------------------------------------------------------------
    def stored_corrections_until_level(self, model_id: int, stop_level: int):
        grid = self.get_grid_net(model_id)
        stored = [self.get_grid_correction(model_id, l) for l in range(grid.num_levels)]
        return [c if l < stop_level else torch.zeros_like(c) for l, c in enumerate(stored)]

    def predict_corrections_until_level(self, model_id: int, stop_level: int, observation: EncoderObservation,
                                        pred_std=0, store_corrections=False):
        """Corrections of levels [0, stop_level) predicted from scratch, the others zero (reference :134-164)."""
        grid = self.get_grid_net(model_id)
        current = [torch.zeros_like(self.get_grid_correction(model_id, l)) for l in range(grid.num_levels)]
        for level in range(stop_level):
            residuals = self.compute_residuals(model_id, current, observation, skip_sign=False, skip_eik=True,
                                               skip_smooth=True)
            inputs = self.compute_encoder_inputs_from_residuals(residuals, model_id, level, save=True)
            outputs = self.compute_encoder_outputs(model_id, inputs, level)
            outputs = outputs + torch.normal(mean=0., std=pred_std, size=outputs.shape).to(outputs)
            assert current[level].shape == outputs.shape
            current[level] = outputs
        if store_corrections:
            with torch.no_grad():
                for level in range(grid.num_levels):
                    self.get_grid_correction(model_id, level).copy_(current[level])
        return current

    def query_sdf(self, model: GridNet, corrections, x):
        """SDF of ``model`` with ``corrections`` added to its grids (differentiable w.r.t. both), reference :166-174."""
        grids = [model.features[l].feature + corrections[l] for l in range(model.num_levels)]
        feats = utils.interp_3d(grids, utils.normalize_coordinates(x, model.bound), second_order_grid_sample=True)
        return utils.grid_decode(feats, x, model.decoder, pos_invariant=True)

    def compute_residuals(self, model_id: int, corrections, observation: EncoderObservation, skip_sign=False,
                          skip_eik=True, skip_smooth=True, smooth_std=0.1):
        """Per-sample constraint values (reference :176-247): SDF error on valid rows; on free-space rows the
        violation of ``sdf <= gt`` and of ``sdf >= trunc_dist``; optionally eikonal and gradient-smoothness terms."""
        x, gt, valid, sign = (observation.coords_world, observation.gt_sdf, observation.gt_sdf_valid,
                              observation.gt_sdf_sign)
        model = self.get_grid_net(model_id)
        pred = self.query_sdf(model, corrections, x)
        zero = torch.zeros_like(pred)
        out = {'sdf_constraint': torch.where(valid == 1, gt - pred, zero), 'sdf_coords': x}
        if not skip_sign:
            upper = torch.where(sign == 1, F.relu(pred - gt), zero)
            lower = torch.where(sign == 1, F.relu(self.trunc_dist - pred), zero)
            out.update(fs_constraint=torch.maximum(upper, lower), fs_upper_constraint=upper, fs_lower_constraint=lower)
        if not skip_eik:
            n = gt.shape[0]
            b = model.bound.detach().cpu().numpy()
            pts = np.concatenate([np.random.uniform(b[a, 0], b[a, 1], n).reshape(n, 1) for a in range(3)], axis=1)
            pts = torch.from_numpy(pts).to(gt).requires_grad_(True)
            s = self.query_sdf(model, corrections, pts)
            with ops.coordinate_gradient_only():
                g = torch.autograd.grad(s, pts, grad_outputs=torch.ones_like(s), create_graph=True)[0]
            out['eik_constraint'] = g.norm(dim=-1) - 1
        if not skip_smooth:
            x1 = x
            x2 = x1 + torch.normal(0, smooth_std, size=x1.shape).to(x1)
            x1.requires_grad_(True)
            x2.requires_grad_(True)
            s1, s2 = self.query_sdf(model, corrections, x1), self.query_sdf(model, corrections, x2)
            with ops.coordinate_gradient_only():
                g1 = torch.autograd.grad(s1, x1, grad_outputs=torch.ones_like(s1), create_graph=True)[0]
                g2 = torch.autograd.grad(s2, x2, grad_outputs=torch.ones_like(s2), create_graph=True)[0]
            out['smooth_constraint'] = torch.where(valid == 1, g1 - g2, torch.zeros_like(g1))
        return out

    def compute_encoder_inputs_from_residuals(self, input_residuals_dict, model_id: int, target_level: int, save=False):
        """(1, 3, Z, Y, X) volume: the three residual signals scatter-averaged onto the target level's cells
        (reference :249-281)."""
        grid = self.get_grid_net(model_id)
        cell = grid.features[target_level].cell_size
        coords = input_residuals_dict['sdf_coords']
        vols = [utils.grid_pool_3d_avg(coords, input_residuals_dict[k], grid.bound, cell).squeeze(-1)
                for k in ('sdf_constraint', 'fs_upper_constraint', 'fs_lower_constraint')]
        inputs = torch.stack(vols, dim=-1).permute(3, 2, 1, 0).unsqueeze(0)
        if save:
            tag = f"model{model_id}_level{target_level}"
            self.intermediate_results[f"encoder_inputs_{tag}"] = inputs.clone().detach()
            self.intermediate_results[f"residuals_coords_{tag}"] = coords.clone().detach()
            self.intermediate_results[f"residuals_values_{tag}"] = input_residuals_dict['sdf_constraint'].clone().detach()
        return inputs

    def compute_encoder_outputs(self, model_id: int, encoder_inputs, target_level: int):
        size = self.get_grid_net(model_id).features[target_level].feature.shape[2:]
        return self.feature_encoders[target_level].predict(None, encoder_inputs, size)


class EncoderPretrainLoss(BaseLoss):
    """Loss for pre-training the per-level predictors (reference :333-400)."""

    def __init__(self, target_level, sdf_weight=3e3, sign_weight=0, eik_weight=0, smooth_weight=0, trunc_dist=0.15,
                 smooth_std=0.01, pred_std=0.1, dataset_groups=[], group_weight=1e3, reg_weight=0):
        super().__init__()
        self.sdf_weight, self.sign_weight, self.eik_weight, self.smooth_weight = (sdf_weight, sign_weight, eik_weight,
                                                                                 smooth_weight)
        self.smooth_std, self.trunc_dist, self.target_level, self.pred_std = smooth_std, trunc_dist, target_level, pred_std
        self.latest_encoder_inputs = {0: {}, 1: {}}
        self.skip_eik = (eik_weight == 0)
        self.skip_smooth = (smooth_weight == 0)

    def compute_loss_from_residuals(self, residuals_dict):
        out = {'sdf': torch.mean(residuals_dict['sdf_constraint'] ** 2) * self.sdf_weight}
        if self.sign_weight > 0:
            out['free_space'] = torch.mean(residuals_dict['fs_constraint']) * self.sign_weight
        if self.eik_weight > 0:
            out['eik'] = torch.mean(residuals_dict['eik_constraint'] ** 2) * self.eik_weight
        if self.smooth_weight > 0:
            out['smooth'] = torch.mean(residuals_dict['smooth_constraint'] ** 2) * self.smooth_weight
        return out

    def compute(self, model: Encoder, model_input: dict, gt: dict) -> dict:
        model_id = model_input['dataset_index'].item()
        grid = model.grid_nets[model.grid_key(model_id)]
        coords_world = utils_geometry.batch_transform_to_world_frame(
            model_input['coords_frame'][0], model_input['frame_indices'][0], model_input['R_world_frame'][0],
            model_input['t_world_frame'][0], grid.rotation_corrections, grid.translation_corrections)
        obs = EncoderObservation(coords_world=coords_world, gt_sdf=gt['sdf'][0], gt_sdf_sign=gt['sdf_signs'][0],
                                 gt_sdf_valid=gt['sdf_valid'][0])
        corrections = model.predict_corrections_until_level(model_id, self.target_level + 1, obs, pred_std=self.pred_std,
                                                            store_corrections=True)
        residuals = model.compute_residuals(model_id, corrections, obs, skip_sign=False, skip_eik=self.skip_eik,
                                            skip_smooth=self.skip_smooth, smooth_std=self.smooth_std)
        loss_dict = self.compute_loss_from_residuals(residuals)
        self.latest_loss_dict = loss_dict
        return loss_dict
