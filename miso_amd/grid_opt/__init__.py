"""Host-side mirror of the reference's ``grid_opt`` package for the hot path.

Same module paths, class names, method names and state-dict keys as the reference
(SURVEY.md 8b), re-implemented on top of ``miso_amd.ops`` (HIP).  Only what the
encode/decode + pose-Jacobian path needs is here; datasets, visualisation, mesh
extraction and the baseline methods are out of scope.

``import miso_amd.compat`` additionally makes this package importable as plain
``grid_opt`` (and ``miso_amd.ops`` as ``cuda_gridsample``), which is what the
reference's demos and pickled atlases expect.
"""
