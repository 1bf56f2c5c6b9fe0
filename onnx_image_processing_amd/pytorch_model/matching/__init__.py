from .sinkhorn import SinkhornMatcher, SinkhornMatcherWithFilters, SinkhornMatcherWithScores

__all__ = ["SinkhornMatcher", "SinkhornMatcherWithScores", "SinkhornMatcherWithFilters"]
