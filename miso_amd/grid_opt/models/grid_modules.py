"""Dense feature grid (reference: grid_opt/models/grid_modules.py:12-123).

The parameter keeps the reference's logical shape (1, C, Z, Y, X) -- state-dicts,
``vertex_positions`` and upstream pickles rely on it -- but is stored
channels-last (``torch.channels_last_3d``): the C features of a voxel are
contiguous, so a trilinear corner is one 16/32-byte access on the GPU instead of
C accesses Z*Y*X floats apart.  Plain-contiguous features (e.g. loaded from an
upstream checkpoint) are accepted by the kernels as well."""
import numpy as np
import torch

import miso_amd.grid_opt.utils.utils as utils
from miso_amd import ops


class FeatureGridBase(torch.nn.Module):
    def __init__(self, d, fdim, bound, cell_size, name="grid", dtype=torch.float32):
        super().__init__()
        assert d in (2, 3)
        assert bound.shape == (d, 2)
        self.d, self.fdim, self.bound = d, fdim, bound
        self.cell_size, self.dtype, self.name = cell_size, dtype, name

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.bound = fn(self.bound)          # plain tensor attribute: follows the module across devices
        return out

    def interpolate(self, x):
        raise NotImplementedError

    def norm(self):
        raise NotImplementedError

    def zero_features(self):
        raise NotImplementedError

    def num_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def lock(self):
        for p in self.parameters():
            p.requires_grad = False

    def unlock(self):
        for p in self.parameters():
            p.requires_grad = True


class FeatureGrid(FeatureGridBase):
    def __init__(self, d, fdim, bound, cell_size, name="grid", dtype=torch.float32,
                 initial_feature=None, init_stddev=0.0, second_order_grid_sample=False):
        super().__init__(d=d, fdim=fdim, bound=bound, cell_size=cell_size, name=name, dtype=dtype)
        if d != 3:
            raise NotImplementedError("the MI355X path covers 3-D grids (every MISO config is 3-D)")
        extent = (bound[:, 1] - bound[:, 0]).cpu().numpy()
        n = np.ceil(extent / cell_size).astype(int)          # voxels along x, y, z
        shape = (1, fdim, int(n[2]), int(n[1]), int(n[0]))   # grid_sample convention: (.., Z, Y, X)
        if initial_feature is None:
            initial_feature = torch.randn(shape, dtype=dtype) * init_stddev
        assert tuple(initial_feature.shape) == shape
        self.feature = torch.nn.Parameter(
            initial_feature.contiguous(memory_format=torch.channels_last_3d))
        # the second-order capable operator is always used; the attribute exists because
        # upstream pickles and callers reference it
        self.grid_sample_func = ops.grid_sample_3d
        self._bound_host = None

    def grid_meta(self, ignore_level=None) -> ops.GridMeta:
        """Host copy of the bound (cached: no device sync per query)."""
        if self.__dict__.get('_bound_host') is None:
            self._bound_host = self.bound.detach().cpu().tolist()
        return ops.GridMeta.from_bound(self._bound_host, ignore_level)

    def interpolate(self, x):
        """(N,3) metres -> (N, fdim), zeros padding, align_corners=False."""
        return ops.encode(x, [self.feature], self.grid_meta())

    def norm(self):
        return self.feature.norm()

    def zero_features(self):
        with torch.no_grad():
            self.feature.zero_()

    def randn_features(self, std):
        with torch.no_grad():
            self.feature.copy_((torch.randn(self.feature.shape, dtype=self.dtype) * std).to(self.feature))

    def vertex_positions(self, denormalize=True) -> torch.Tensor:
        """Centres of all voxels, (Z*Y*X, 3), z-major, metres by default."""
        pos = utils.all_grid_positions(self.feature).reshape(-1, 3)
        if denormalize:
            return utils.denormalize_coordinates(pos, self.bound.to(pos))
        return pos
