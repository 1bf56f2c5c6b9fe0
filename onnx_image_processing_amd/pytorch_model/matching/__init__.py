from . import outlier_filters
from .sinkhorn import SinkhornMatcher, SinkhornMatcherWithFilters, SinkhornMatcherWithScores

__all__ = ["SinkhornMatcher", "SinkhornMatcherWithScores", "SinkhornMatcherWithFilters", "outlier_filters"]
