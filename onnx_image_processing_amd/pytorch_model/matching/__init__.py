from .sinkhorn import SinkhornMatcher, SinkhornMatcherWithScores

__all__ = ["SinkhornMatcher", "SinkhornMatcherWithScores"]
