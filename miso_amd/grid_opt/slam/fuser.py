"""Fuser: align the submaps of an atlas (reference: grid_opt/slam/fuser.py:12-54; the
upstream ``fuse`` method is broken -- it passes a kwarg its loss does not accept -- and is
not reproduced)."""
import math

from torch.utils.data import DataLoader
from miso_amd.grid_opt.utils.utils import collate_batch_of_one

from miso_amd.grid_opt.align.miso import align_multiple_submaps_hierarchical
from miso_amd.grid_opt.models.grid_atlas import GridAtlas


class Fuser:
    def __init__(self, model: GridAtlas, dataset, cfg: dict):
        assert isinstance(model, GridAtlas), "Model must be an instance of GridAtlas."
        self.model = model
        self.dataset = dataset
        self.train_loader = DataLoader(dataset, shuffle=True, batch_size=1, num_workers=0, collate_fn=collate_batch_of_one)
        self.cfg = cfg

    def align(self):
        a = self.cfg['align']
        info = align_multiple_submaps_hierarchical(
            grid_atlas=self.model, dataset=self.dataset, level_iters=a['level_iters'],
            finetune_iters=a['finetune_iters'], level_thresh=0, lr=a['learning_rate'],
            align_loss=a['loss_type'], stability_thresh=a['stability_thresh'],
            subsample_points=a['subsample_points'], latent_levels=a['latent_levels'],
            skip_finetune=a['skip_finetune'], pose_reg_weight=a['pose_reg_weight'],
            pose_thresh_m=a.get('pose_thresh_m', 10.0),
            pose_thresh_rad=math.radians(a.get('pose_thresh_deg', 45.0)),
            verbose=a.get('verbose', False), save_iterations=a.get('save_iterations', False),
            device=self.cfg.get('device', 'cuda:0'))
        self.model.print_submap_pose_info()
        return info
