"""SinkhornMatcher -- mirror of reference pytorch_model/matching/sinkhorn.py:28-259."""
import torch
from torch import nn

from ... import _native as N
from ... import ops


class SinkhornSolution:
    """Result of a Sinkhorn solve kept as duals: P = exp(Z + u + v) is defined by the log-score
    source (fp32 Z, or uint16 dot products + per-descriptor scale/norm) and the duals u, v, all
    resident on the GPU.  `mutual_matches` is MutualNearestNeighborMatcher.forward on that P
    without writing it (bit-identical to forward() + the extractor)."""

    def __init__(self, kind: str, source, m: int, pitch: int, epsilon: float, u: torch.Tensor, v: torch.Tensor):
        self.kind, self.source, self.m, self.pitch, self.epsilon, self.u, self.v = kind, source, m, pitch, epsilon, u, v

    def mutual_matches(self, keypoints1: torch.Tensor, keypoints2: torch.Tensor, max_matches: int, threshold: float,
                       return_indices: bool = False):
        if self.kind == "dots":
            return ops.mnn_from_duals_dots(self.source, self.m, self.epsilon, self.u, self.v, keypoints1, keypoints2,
                                           max_matches, threshold, return_indices)
        return ops.mnn_from_duals(self.source, self.m, self.pitch, self.u, self.v, keypoints1, keypoints2,
                                  max_matches, threshold, return_indices)


class SinkhornMatcher(nn.Module):
    """Log-space Sinkhorn assignment with dustbins.

    forward(desc1 (B,N,D), desc2 (B,M,D)) -> P (B,N+1,M+1) float32.  Arguments and validation
    follow sinkhorn.py:57-77.  K5 builds the n x m log-score core with MFMA, K6 runs the
    fixed number of row/column log-sum-exp passes; the constant dustbin row/column is applied
    analytically rather than stored.
    """

    def __init__(self, iterations: int = 20, epsilon: float = 1.0, unused_score: float = 1.0,
                 distance_type: str = "l2") -> None:
        super().__init__()
        if iterations <= 0:
            raise ValueError(f"iterations must be positive, got {iterations}")
        if epsilon <= 0:
            raise ValueError(f"epsilon must be positive, got {epsilon}")
        self.iterations = iterations
        self.epsilon = epsilon
        self.unused_score = unused_score
        self.distance_type = distance_type.lower()
        if self.distance_type not in ("l1", "l2"):
            raise ValueError(f"distance_type must be 'l1' or 'l2', got {distance_type}")
        # forward_bits / solve_bits: keep the K x K dot products as uint16 and rebuild Z in registers
        # every pass (half the bytes per iteration; the fp32-Z iteration is HBM-bound on MI355X).
        # Applies to M <= 1024; larger problems use the fp32-Z form.
        self.use_dot_storage = True

    @property
    def dustbin_logscore(self) -> float:
        return -self.unused_score / self.epsilon            # sinkhorn.py:182 (Python-float arithmetic)

    @torch.no_grad()
    def forward(self, desc1: torch.Tensor, desc2: torch.Tensor) -> torch.Tensor:
        dist = N.MI_DIST_L2 if self.distance_type == "l2" else N.MI_DIST_L1
        z, pitch = ops.cost_logscores_f32(desc1, desc2, dist, self.epsilon)
        return ops.sinkhorn(z, desc2.shape[1], pitch, self.dustbin_logscore, self.iterations)

    @torch.no_grad()
    def forward_bits(self, bits1: torch.Tensor, bits2: torch.Tensor, normalized: bool) -> torch.Tensor:
        """Same result for hard-binarised descriptors given as packed bits (B,N,D/32) int32:
        the dot products become exact integer popcounts (i8 MFMA).  L2 only."""
        if self.distance_type != "l2":
            raise RuntimeError("forward_bits implements the l2 cost only")
        if self.use_dot_storage and ops.dots_supported(bits1.shape[0], bits1.shape[1], bits2.shape[1], self.epsilon):
            return ops.sinkhorn_bits(bits1, bits2, normalized, self.epsilon, self.unused_score, self.iterations)
        z, pitch = ops.cost_logscores_bits(bits1, bits2, normalized, self.epsilon)
        return ops.sinkhorn(z, bits2.shape[1], pitch, self.dustbin_logscore, self.iterations)

    @torch.no_grad()
    def solve(self, desc1: torch.Tensor, desc2: torch.Tensor) -> SinkhornSolution:
        """forward() without the final exp: the duals (extension; see SinkhornSolution)."""
        dist = N.MI_DIST_L2 if self.distance_type == "l2" else N.MI_DIST_L1
        z, pitch = ops.cost_logscores_f32(desc1, desc2, dist, self.epsilon)
        _, u, v = ops.sinkhorn(z, desc2.shape[1], pitch, self.dustbin_logscore, self.iterations, want_p=False)
        return SinkhornSolution("z", z, desc2.shape[1], pitch, self.epsilon, u, v)

    @torch.no_grad()
    def solve_bits(self, bits1: torch.Tensor, bits2: torch.Tensor, normalized: bool) -> SinkhornSolution:
        if self.distance_type != "l2":
            raise RuntimeError("solve_bits implements the l2 cost only")
        m = bits2.shape[1]
        if self.use_dot_storage and ops.dots_supported(bits1.shape[0], bits1.shape[1], m, self.epsilon):
            _, u, v, state = ops.sinkhorn_bits(bits1, bits2, normalized, self.epsilon, self.unused_score,
                                               self.iterations, want_p=False, return_state=True)
            return SinkhornSolution("dots", state, m, state[3], self.epsilon, u, v)
        z, pitch = ops.cost_logscores_bits(bits1, bits2, normalized, self.epsilon)
        _, u, v = ops.sinkhorn(z, m, pitch, self.dustbin_logscore, self.iterations, want_p=False)
        return SinkhornSolution("z", z, m, pitch, self.epsilon, u, v)


class SinkhornMatcherWithScores(SinkhornMatcher):
    """forward -> (P, scores0 (B,N), scores1 (B,M)): row / column maxima of the core of P
    (sinkhorn.py:228-259)."""

    @torch.no_grad()
    def forward(self, desc1: torch.Tensor, desc2: torch.Tensor):
        p = super().forward(desc1, desc2)
        scores0, scores1 = ops.core_maxima(p)
        return p, scores0, scores1


class SinkhornMatcherWithFilters(SinkhornMatcher):
    """forward -> (P_filtered (B,N+1,M+1), valid_mask (B,N) bool): Sinkhorn followed by the
    probability-ratio filter (best / (second + 1e-8) >= ratio_threshold) and the dustbin-margin
    filter (best - P[i, M] >= dustbin_margin); rows that fail are reassigned to the dustbin
    (sinkhorn.py:303-315 arguments, :391-465 forward).  None disables a filter, as in the
    reference.  K7 `mi_match_filters` works in place on the freshly computed P."""

    def __init__(self, iterations: int = 20, epsilon: float = 1.0, unused_score: float = 1.0,
                 distance_type: str = "l2", ratio_threshold: float = None, dustbin_margin: float = None) -> None:
        super().__init__(iterations, epsilon, unused_score, distance_type)
        self.ratio_threshold = ratio_threshold if ratio_threshold is not None else -1.0
        self.dustbin_margin = dustbin_margin if dustbin_margin is not None else -1.0

    @torch.no_grad()
    def forward(self, desc1: torch.Tensor, desc2: torch.Tensor):
        return ops.match_filters(super().forward(desc1, desc2), self.ratio_threshold, self.dustbin_margin)

    @torch.no_grad()
    def forward_bits(self, bits1: torch.Tensor, bits2: torch.Tensor, normalized: bool):
        return ops.match_filters(super().forward_bits(bits1, bits2, normalized), self.ratio_threshold,
                                 self.dustbin_margin)
