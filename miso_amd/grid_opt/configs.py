"""YAML config loading with ``inherit_from`` + recursive default merge, and the
factories the hot path needs (reference: grid_opt/configs.py).  Dataset factories are
out of scope (disk I/O + sampling)."""
import yaml
import torch
from torch.utils.data import DataLoader  # noqa: F401  (the reference's demos star-import it from here)

from .loss import *        # noqa: F401,F403
from .loss_isdf import iSDFLoss, iSDFLossSubmap
from .models.grid_net import GridNet
from .trainer import *     # noqa: F401,F403
from .trainer import GridTrainer, Trainer


def update_recursive(dict1, dict2):
    for k, v in dict2.items():
        if k not in dict1:
            dict1[k] = dict()
        if isinstance(v, dict):
            update_recursive(dict1[k], v)
        else:
            dict1[k] = v


def load_config(path, default_path=None):
    with open(path, 'r') as f:
        special = yaml.full_load(f)
    parent = special.get('inherit_from')
    if parent is not None:
        cfg = load_config(parent, default_path)
    elif default_path is not None:
        with open(default_path, 'r') as f:
            cfg = yaml.full_load(f)
    else:
        cfg = dict()
    update_recursive(cfg, special)
    return cfg


def cfg_model(cfg):
    name = cfg['model']['name']
    if name != 'grid_net':
        raise ValueError(f"model {name!r} is outside the MI355X hot path (grid_net only)")
    model = GridNet(cfg=cfg['model'], device=cfg['device'], dtype=torch.float32)
    model.to(cfg['device'])
    return model


_ISDF_KEYS = ('trunc_weight', 'trunc_distance', 'noise_std', 'eik_weight', 'grad_weight', 'eik_apply_dist',
              'smooth_weight', 'smooth_std', 'loss_type', 'slam_mode', 'pose_reg_weight', 'pose_thresh_m',
              'pose_thresh_rad')


def cfg_loss(cfg):
    name = cfg['loss']['name']
    if name in ('iSDF', 'iSDFSubmap'):
        kw = {k: cfg['loss'][k] for k in _ISDF_KEYS}
        kw.update(model_name=cfg['model']['name'], orien_loss=bool(cfg['loss']['orien_loss']))
        if name == 'iSDF':
            return iSDFLoss(**kw)
        return iSDFLossSubmap(feat_reg_weight=1.0, **kw)
    raise ValueError(f"loss {name!r} is outside the MI355X hot path (iSDF / iSDFSubmap; the Miso* losses "
                     "are constructed directly by Mapper / Tracker)")


def cfg_trainer(cfg, model, train_loader, val_loader=None):
    kind = cfg['train'].setdefault('trainer', 'base')
    if kind not in ('base', 'grid'):
        raise ValueError(f"Invalid trainer type: {kind}.")
    cls = Trainer if kind == 'base' else GridTrainer
    return cls(cfg['train'], model, cfg_loss(cfg), train_loader, val_loader, cfg['device'], torch.float32)
