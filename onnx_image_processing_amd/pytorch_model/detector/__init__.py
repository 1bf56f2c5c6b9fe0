from .akaze import AKAZE, HessianDetector, NonLinearDiffusion, OrientationEstimator
from .shi_tomasi import ShiTomasiScore

__all__ = ["AKAZE", "HessianDetector", "NonLinearDiffusion", "OrientationEstimator", "ShiTomasiScore"]
