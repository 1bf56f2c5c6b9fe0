from .akaze_sparse_bad_sinkhorn import AKAZESparseBADSinkhornMatcher
from .essential_matrix import (AKAZESparseBADSinkhornWithEssentialMatrix,
                               ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix)
from .match_extraction_wrapper import MatchExtractionWrapper
from .shi_tomasi_angle import ShiTomasiAngleSparseBAD, ShiTomasiAngleSparseBADDetector, ShiTomasiWithAngle
from .shi_tomasi_angle_sparse_bad_sinkhorn import (ShiTomasiAngleSparseBADSinkhornMatcher,
                                                   ShiTomasiAngleSparseBADSinkhornMatcherWithFilters)
from .shi_tomasi_bad import ShiTomasiBADDetector
from .shi_tomasi_bad_sinkhorn import ShiTomasiBADSinkhornMatcher
from .shi_tomasi_sparse_bad_sinkhorn import ShiTomasiSparseBADSinkhornMatcher

__all__ = ["AKAZESparseBADSinkhornMatcher", "AKAZESparseBADSinkhornWithEssentialMatrix",
           "ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix", "ShiTomasiBADDetector", "ShiTomasiBADSinkhornMatcher", "ShiTomasiSparseBADSinkhornMatcher", "MatchExtractionWrapper", "ShiTomasiWithAngle",
           "ShiTomasiAngleSparseBAD", "ShiTomasiAngleSparseBADDetector", "ShiTomasiAngleSparseBADSinkhornMatcher",
           "ShiTomasiAngleSparseBADSinkhornMatcherWithFilters"]
