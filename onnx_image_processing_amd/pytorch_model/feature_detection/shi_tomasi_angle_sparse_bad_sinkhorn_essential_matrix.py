"""Module name of reference pytorch_model/feature_detection/shi_tomasi_angle_sparse_bad_sinkhorn_essential_matrix.py
(:34-361); the class lives in essential_matrix.py next to its AKAZE sibling (they share the tail)."""
from .essential_matrix import ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix

__all__ = ["ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix"]
