from .angle_estimation import AngleEstimator

__all__ = ["AngleEstimator"]
