"""Module name of reference pytorch_model/feature_detection/akaze_sparse_bad_sinkhorn_essential_matrix.py (:34-378);
the class lives in essential_matrix.py next to its Shi-Tomasi sibling (they share the tail)."""
from .essential_matrix import AKAZESparseBADSinkhornWithEssentialMatrix

__all__ = ["AKAZESparseBADSinkhornWithEssentialMatrix"]
