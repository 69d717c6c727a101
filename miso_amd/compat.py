"""Drop-in import names of the reference.

``import miso_amd.compat`` registers
  * ``grid_opt`` (and every ``grid_opt.*`` submodule) as aliases of ``miso_amd.grid_opt``,
    the dotted paths the reference's demos import (demo/build_submaps.py:4-11) and its
    pickled atlases name (``torch.save(grid_atlas, ...)``, demo/build_submaps.py:141);
  * ``cuda_gridsample`` with ``grid_sample_2d/3d`` (third_party/cuda_gridsample_grad2/
    cuda_gridsample.py:12-19), which FeatureGrid pickles reference by name.
Aliases resolve to the SAME module objects (no double import).
"""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types

_PREFIX = "grid_opt"
_TARGET = "miso_amd.grid_opt"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _PREFIX or fullname.startswith(_PREFIX + "."):
            real = _TARGET + fullname[len(_PREFIX):]
            try:
                importlib.import_module(real)
            except ModuleNotFoundError:
                return None
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=True)
        return None


def install():
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())
    if "cuda_gridsample" not in sys.modules:
        from . import ops
        mod = types.ModuleType("cuda_gridsample")
        mod.grid_sample_3d = ops.grid_sample_3d

        def grid_sample_2d(input, grid, padding_mode='zeros', align_corners=True):
            """2-D sampling through the 3-D operator: a (1,C,H,W) image is a depth-1 volume
            sampled at z = 0."""
            import torch
            assert input.ndim == 4 and grid.ndim == 4 and grid.shape[3] == 2
            g3 = torch.cat((grid, torch.zeros_like(grid[..., :1])), dim=-1).unsqueeze(1)
            out = ops.grid_sample_3d(input.unsqueeze(2), g3, padding_mode=padding_mode,
                                     align_corners=align_corners)
            return out.squeeze(2)

        mod.grid_sample_2d = grid_sample_2d
        sys.modules["cuda_gridsample"] = mod


install()
