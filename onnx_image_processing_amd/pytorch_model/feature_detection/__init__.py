from .match_extraction_wrapper import MatchExtractionWrapper
from .shi_tomasi_sparse_bad_sinkhorn import ShiTomasiSparseBADSinkhornMatcher

__all__ = ["ShiTomasiSparseBADSinkhornMatcher", "MatchExtractionWrapper"]
