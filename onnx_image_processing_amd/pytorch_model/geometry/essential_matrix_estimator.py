"""EssentialMatrixEstimator -- mirror of reference
pytorch_model/geometry/essential_matrix_estimator.py:29-431."""
import torch
from torch import nn

from ... import ops


class EssentialMatrixEstimator(nn.Module):
    """forward(P (N+1, M+1)) -> E (3, 3): weighted 8-point algorithm on a Sinkhorn assignment matrix
    whose feature i sits at pixel (i % W, i // W) of an `image_shape` grid.  Constructor arguments
    and buffers (`K`, `K_inv`, `pixel_coords`, `pixel_coords_n`) follow :73-118; the arithmetic
    (bidirectional top-k mask, Hartley normalisation, normal equations, shifted power iteration,
    manifold projection) runs in K10 `mi_essential_matrix`.  A batched P (B, N+1, M+1) gives
    (B, 3, 3) (extension).  `estimate(P, pts1_n, pts2_n, valid1, valid2)` is what the composites'
    `_estimate_essential_matrix` computes from actual keypoints."""

    def __init__(self, K: torch.Tensor, image_shape: tuple[int, int] = (32, 32), top_k: int = 3, n_iter: int = 30,
                 n_iter_manifold: int = 10) -> None:
        super().__init__()
        K_f = K.float()
        self.register_buffer("K", K_f)
        self.register_buffer("K_inv", torch.linalg.inv(K_f.cpu()).to(K_f.device))
        self.top_k = top_k
        self.n_iter = n_iter
        self.n_iter_manifold = n_iter_manifold
        H, W = image_shape
        self.H = H
        self.W = W
        idx = torch.arange(H * W, dtype=torch.float32)
        pixel_coords = torch.stack([idx % W, idx // W], dim=-1)                       # (H*W, 2) as (x, y)
        self.register_buffer("pixel_coords", pixel_coords)
        hom = torch.cat([pixel_coords, torch.ones(H * W, 1)], dim=-1)
        self.register_buffer("pixel_coords_n", (hom @ self.K_inv.cpu().T)[:, :2].contiguous())

    @torch.no_grad()
    def estimate(self, P: torch.Tensor, pts1_n: torch.Tensor, pts2_n: torch.Tensor, valid1=None, valid2=None):
        """P (B,N+1,M+1), normalised (x,y) points (B,N,2)/(B,M,2), optional validity masks -> E (B,3,3)."""
        return ops.essential_matrix(P, pts1_n, pts2_n, valid1, valid2, self.top_k, self.n_iter, self.n_iter_manifold)

    @torch.no_grad()
    def estimate_from_solution(self, sol, pts1_n: torch.Tensor, pts2_n: torch.Tensor, valid1=None, valid2=None):
        """estimate() on the P a packed-form SinkhornSolution defines (matching/sinkhorn.py: kind "dots"), without that P
        being written (`mi_essential_matrix_dots`): same E bit for bit."""
        if sol.kind != "dots":
            raise RuntimeError("estimate_from_solution needs the packed dot-product form of the Sinkhorn solution")
        return ops.essential_matrix_dots(sol.source, sol.m, sol.epsilon, sol.u, sol.v, pts1_n, pts2_n, valid1, valid2,
                                         self.top_k, self.n_iter, self.n_iter_manifold)

    @torch.no_grad()
    def forward(self, P: torch.Tensor) -> torch.Tensor:
        single = P.dim() == 2
        pb = P.unsqueeze(0) if single else P
        b, n, m = pb.shape[0], pb.shape[1] - 1, pb.shape[2] - 1
        if max(n, m) > self.H * self.W:
            raise RuntimeError(f"image_shape {self.H}x{self.W} holds fewer than max(N, M) = {max(n, m)} grid points")
        grid = self.pixel_coords_n.to(pb.device)
        e = self.estimate(pb, grid[:n].expand(b, n, 2), grid[:m].expand(b, m, 2))
        return e[0] if single else e
