from .so3 import hat, so3_exp_map


def so3_relative_angle(*a, **k):
    raise NotImplementedError("shim: not on the hot path")


def matrix_to_axis_angle(*a, **k):
    raise NotImplementedError("shim: not on the hot path")
