"""miso_amd -- MI355X-native implementation of MISO's encode/decode hot path.

``miso_amd.ops``       autograd operators over the C ABI of libmiso_hip.so
``miso_amd.grid_opt``  host-side mirror of the reference's ``grid_opt`` API
                       (FeatureGrid, GridNet, GridAtlas, losses, trainer, align)
``miso_amd.dist``      submap-parallel sharding over torch.distributed (RCCL)

The HIP library is the product path; nothing here falls back to CPU compute.
"""
__version__ = "0.1.0"
