"""Camera description used by the RGB-D dataset (reference: grid_opt/utils/utils_data.py:7-15)."""
from dataclasses import dataclass


@dataclass
class CameraParameters:
    fx: float
    fy: float
    cx: float
    cy: float
    H: int
    W: int
    depth_scale: float = 1000.0
