from .bad import SparseBAD

__all__ = ["SparseBAD"]
