"""Common base of the scene models (reference: grid_opt/models/base_net.py)."""
import torch
import torch.nn as nn


class BaseNet(nn.Module):
    def __init__(self, cfg: dict, device='cpu', dtype=torch.float32):
        super().__init__()
        self.cfg = cfg
        self.d = cfg['spatial_dim']
        assert self.d in (2, 3)
        self.device = device
        self.dtype = dtype
        self.bound = torch.tensor(cfg['grid']['bound'], device=device, dtype=dtype)
        assert self.bound.shape == (self.d, 2)

    def _apply(self, fn, *args, **kwargs):
        """``.to()`` / ``.cuda()`` also move the bound, a plain tensor attribute (upstream leaves it on the device
        the model was built on)."""
        out = super()._apply(fn, *args, **kwargs)
        self.bound = fn(self.bound)
        if isinstance(self.device, (str, torch.device)):
            self.device = self.bound.device
        return out

    # device-side caches (ctypes structs with raw pointers, pose tables, vertex tables) are rebuilt on demand and must
    # not travel with ``torch.save(model)`` -- the whole-module pickle the demos exchange (demo/build_submaps.py:141)
    _TRANSIENT = ('_pose_cache', '_vertex_cache', '_align_src_cache', '_align_grid_cache', '_align_plan_const',
                  '_kf_pose_cache', '_kf_table', '_fast_plans', '_atlas_query', '_atlas_poses', '_atlas_eligible')

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._TRANSIENT:
            state.pop(k, None)
        return state

    def forward(self, x: torch.Tensor):
        raise NotImplementedError

    def params_at_level(self, level):
        raise NotImplementedError

    def print_trainable_params(self):
        print("\n === Trainable parameters === ")
        for name, param in self.named_parameters():
            if param.requires_grad:
                print(f"{name}: {param.shape}")
        print("=== END trainable parameters === \n ")
