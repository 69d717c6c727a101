"""Decoder MLP (reference: grid_opt/models/modules.py:11-40).  State-dict keys
``network.{0,2,4,..}.{weight,bias}`` are kept so upstream decoder checkpoints load."""
import torch
from torch import nn

from miso_amd import ops


class MLPNet(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim=64, hidden_layers=1, bias=False,
                 acti_func=nn.ReLU, pretrained_path=None, no_optimize=False):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        dims = [input_dim] + [hidden_dim] * (hidden_layers + 1)
        self.layers = []
        for a, b in zip(dims[:-1], dims[1:]):
            self.layers += [nn.Linear(a, b, bias=bias), acti_func()]
        self.layers.append(nn.Linear(hidden_dim, output_dim, bias=bias))
        self.network = nn.Sequential(*self.layers)
        self._relu_only = acti_func is nn.ReLU
        if pretrained_path is not None:
            self.load(pretrained_path)
        if no_optimize:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        return self.network(x)

    def save(self, filepath):
        torch.save(self.state_dict(), filepath)

    def load(self, filepath):
        self.load_state_dict(torch.load(filepath))

    # -- glue to the fused HIP path ------------------------------------------------
    def linears(self):
        lin = self.__dict__.get('_linears')
        if lin is None:        # the layer set is fixed at construction; walking nn.Module containers costs ~20 us a call
            lin = self.__dict__['_linears'] = [m for m in self.network if isinstance(m, nn.Linear)]
        return lin

    def is_frozen(self) -> bool:
        ps = self.__dict__.get('_param_list')
        if ps is None:
            ps = self.__dict__['_param_list'] = list(self.parameters())
        return not any(p.requires_grad for p in ps)

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ('_pack', '_linears', '_param_list'):      # packed weights (ctypes struct + device buffer) and the
            state.pop(k, None)                             # cached layer / parameter lists: rebuilt on demand
        return state

    def decoder_pack(self):
        """Packed weights for the fused encode+decode kernels, or None if the decoder
        cannot take that path (trainable weights, non-ReLU activation)."""
        relu_only = self.__dict__.get('_relu_only')
        if relu_only is None:
            # a module unpickled from a file the reference wrote (torch.save(grid_atlas), demo/build_submaps.py:141) carries
            # the reference's attributes only: read the activation off the layer list
            relu_only = self.__dict__['_relu_only'] = all(
                isinstance(m, (nn.Linear, nn.ReLU)) for m in self.network)
        if not relu_only or not self.is_frozen():
            return None
        pack = self.__dict__.get('_pack')
        lin = self.linears()
        ws, bs = [l.weight for l in lin], [l.bias for l in lin]
        if pack is None or any(a is not b for a, b in zip(pack.weights, ws)):
            pack = ops.DecoderPack(ws, bs)
            self.__dict__['_pack'] = pack   # not a module attribute: stays out of state_dict / pickles
        return pack


# --------------------------------------------------------------------------- #
# Learned initialisation: the per-level feature predictor
# --------------------------------------------------------------------------- #
class ConvInterp(nn.Module):
    """A few 3x3x3 convolutions (+ReLU, optional max-pool after each) whose output volume is resampled to a
    requested spatial size: trilinear when growing, area-averaged when shrinking (reference modules.py:107-181).
    The convolutions are plain library calls (MIOpen) -- they see one small residual volume per level."""

    def __init__(self, dim, in_channels, base_channels=4, hidden_layers=2, kernel_size=3, padding=1,
                 reduction_factor=2, dtype=torch.float32, device='cuda:0', name='ConvInterp'):
        super().__init__()
        if dim != 3:
            raise ValueError("only the 3-D variant is on the MISO path (2-D grids are unused by every config)")
        self.name, self.d = name, dim
        self.pool = nn.MaxPool3d(kernel_size=reduction_factor, stride=reduction_factor) if reduction_factor > 1 else None
        self.conv_layers = nn.ModuleList()
        width = in_channels
        for i in range(hidden_layers):
            out = base_channels * (2 ** i)
            self.conv_layers.append(nn.Conv3d(width, out, kernel_size=kernel_size, stride=1, padding=padding,
                                              dtype=dtype, device=device))
            width = out
        self.output_channels = width

    def forward(self, x):
        for conv in self.conv_layers:
            x = torch.relu(conv(x))
            if self.pool is not None:
                x = self.pool(x)
        return x

    def forward_and_interpolate(self, x, output_spatial_size):
        x = self.forward(x)
        have, want = tuple(x.shape[2:]), tuple(int(v) for v in output_spatial_size)
        if all(h <= w for h, w in zip(have, want)):
            return nn.functional.interpolate(x, size=want, mode='trilinear', align_corners=False)
        if all(h > w for h, w in zip(have, want)):
            return nn.functional.interpolate(x, size=want, mode='area')
        raise ValueError(f"Invalid input and output size! input_size={have}, output_size={want}. ")


class FeaturePrediction(nn.Module):
    """Per-voxel features of one grid level from pooled residual volumes (and optionally coarser features): two
    ConvInterp towers resampled to the level's shape, concatenated per voxel, then a 2x16 MLP with bias
    (reference modules.py:235-319).  State-dict keys as upstream: ``residual_processor.conv_layers.*``,
    ``feature_processor.conv_layers.*``, ``mlp.network.*``."""

    def __init__(self, d, fdim, rdim=1, feature_processor=True, residual_processor=True, normalize_output=False,
                 device='cuda:0', initial_param_std=None):
        super().__init__()
        self.d = d
        width = 0
        self.feature_processor = self.residual_processor = None
        if feature_processor:
            self.feature_processor = ConvInterp(d, fdim, reduction_factor=1, hidden_layers=2, device=device,
                                                name=f'feature_proc_{d}D')
            width += self.feature_processor.output_channels
        if residual_processor:
            self.residual_processor = ConvInterp(d, rdim, reduction_factor=1, hidden_layers=2, device=device,
                                                 name=f'residual_proc_{d}D')
            width += self.residual_processor.output_channels
        self.mlp = MLPNet(input_dim=width, output_dim=fdim, hidden_dim=16, hidden_layers=2, bias=True).to(device)
        self.normalize_output = normalize_output
        if initial_param_std is not None:
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, std=initial_param_std)
                    nn.init.normal_(m.bias, std=initial_param_std)

    def predict(self, coarse_features, coarse_residuals, output_spatial_size):
        """(1, fdim, *size): coarse_features (1,fdim,..) or None, coarse_residuals (1,rdim,..) or None."""
        if self.d != 3:
            raise ValueError(f"Invalid dimension: {self.d}!")
        towers = []
        for proc, vol in ((self.feature_processor, coarse_features), (self.residual_processor, coarse_residuals)):
            if proc is not None:
                towers.append(proc.forward_and_interpolate(vol, output_spatial_size)[0])        # (C, H, W, D)
        assert towers, "Input to MLP is empty! "
        per_voxel = torch.cat(towers, dim=0).flatten(1).T                                       # (H*W*D, C)
        out = self.mlp(per_voxel)
        if self.normalize_output:
            out = out / (out.norm(dim=-1, keepdim=True) + 1e-8)          # utils.normalize_last_dim
        size = tuple(int(v) for v in output_spatial_size)
        return out.T.reshape(1, -1, *size)
