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
        return [m for m in self.network if isinstance(m, nn.Linear)]

    def is_frozen(self) -> bool:
        return not any(p.requires_grad for p in self.parameters())

    def decoder_pack(self):
        """Packed weights for the fused encode+decode kernels, or None if the decoder
        cannot take that path (trainable weights, non-ReLU activation)."""
        if not self._relu_only or not self.is_frozen():
            return None
        pack = self.__dict__.get('_pack')
        lin = self.linears()
        ws, bs = [l.weight for l in lin], [l.bias for l in lin]
        if pack is None or any(a is not b for a, b in zip(pack.weights, ws)):
            pack = ops.DecoderPack(ws, bs)
            self.__dict__['_pack'] = pack   # not a module attribute: stays out of state_dict / pickles
        return pack
