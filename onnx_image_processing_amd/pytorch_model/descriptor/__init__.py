from .bad import BADDescriptor, SparseBAD

__all__ = ["BADDescriptor", "SparseBAD"]
