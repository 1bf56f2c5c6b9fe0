from .essential_matrix_estimator import EssentialMatrixEstimator

__all__ = ["EssentialMatrixEstimator"]
