from .akaze import AKAZE, HessianDetector, NonLinearDiffusion, OrientationEstimator
from .dog import DoGDetector, DoGDetectorWithScore
from .fast import FASTScore
from .shi_tomasi import ShiTomasiScore

__all__ = ["AKAZE", "DoGDetector", "DoGDetectorWithScore", "FASTScore", "HessianDetector", "NonLinearDiffusion",
           "OrientationEstimator", "ShiTomasiScore"]
