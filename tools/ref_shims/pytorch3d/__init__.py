"""Import shim used ONLY by tools/make_goldens.py to import the reference on CPU.

pytorch3d is an un-vendored, unpinned dependency of the reference
(environment.yaml:114) and is absent from this image.  Only so3_exp_map / hat
carry arithmetic the hot path needs; they are restated here from pytorch3d's
published formula (parity unpinned at this boundary, see oracle/ref_torch.py).
"""
