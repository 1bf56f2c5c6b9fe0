"""torch-CPU restatement of the reference hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` may import this module.  It exists because the
reference's Python files cannot travel to the GPU box, while the benchmark has to time "the reference CPU path" on
that box's host cores: the same ATen CPU kernels in the same order as
the reference's modules (conv2d, max_pool2d, topk, grid_sample, bmm, logsumexp), restated here from the cited lines.
It is pinned in the build container against the fixtures the imported reference produced
(tests/test_oracle_golden.py::test_torch_cpu_restatement_*: keypoints and BAD bits exact, P to 1e-6), so what it
times is what the reference would do.  The numpy oracle (numpy_oracle.py) stays the exact-arithmetic checker;
this file is the *timing* twin.

Citations are paths under the reference root.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _sobel_pair() -> torch.Tensor:
    # detector/shi_tomasi.py:47-59: unnormalised 3x3 Sobel, x then y, cross-correlation
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
    return torch.stack([kx, kx.t()]).unsqueeze(1)


class TorchCpuPath:
    """ShiTomasi -> NMS/top-k -> SparseBAD (nearest, non-oriented) -> Sinkhorn -> mutual NN, the composition of
    feature_detection/shi_tomasi_sparse_bad_sinkhorn.py:156-180 + match_extraction_wrapper.py:82-113."""

    def __init__(self, box: np.ndarray, thr: np.ndarray, max_keypoints: int, block_size: int = 3, binarize: bool = True,
                 soft_binarize: bool = False, temperature: float = 10.0, sinkhorn_iterations: int = 20,
                 epsilon: float = 0.05, unused_score: float = 1.0, nms_radius: int = 5, score_threshold: float = 0.0,
                 normalize_descriptors: bool = True, border_margin: int | None = None):
        self.k = int(max_keypoints)
        self.bs = int(block_size)
        self.binarize, self.soft, self.temperature = binarize, soft_binarize, float(temperature)
        self.iters, self.eps, self.unused = int(sinkhorn_iterations), float(epsilon), float(unused_score)
        self.r_nms, self.thr_score, self.normalize = int(nms_radius), float(score_threshold), normalize_descriptors
        self.sobel = _sobel_pair()
        self.box_sum = torch.ones(3, 1, self.bs, self.bs)                      # shi_tomasi.py:63-64
        t = torch.from_numpy(np.asarray(box, np.float32))
        self.off = [(t[:, c] - 16.0).view(1, 1, -1) for c in range(4)]           # x1, x2, y1, y2 (bad.py:403-415)
        radii = t[:, 4].long()
        self.thr = torch.from_numpy(np.asarray(thr, np.float32)).view(1, 1, -1)
        self.rmax = int(radii.max())
        self.margin = self.rmax if border_margin is None else int(border_margin)
        sel = torch.zeros(self.rmax + 1, t.shape[0])
        sel[radii, torch.arange(t.shape[0])] = 1.0                               # bad.py:420-423
        self.sel = sel.view(1, self.rmax + 1, 1, -1)
        r = torch.arange(self.rmax + 1, dtype=torch.float32).view(-1, 1, 1)
        c = torch.arange(-self.rmax, self.rmax + 1, dtype=torch.float32)
        inside = ((c.abs().view(1, -1, 1) <= r) & (c.abs().view(1, 1, -1) <= r)).float()
        self.bank = (inside / (2.0 * r + 1.0) ** 2).unsqueeze(1)                # bad.py:426-434

    # detector/shi_tomasi.py:78-110
    def scores(self, img: torch.Tensor) -> torch.Tensor:
        g = F.conv2d(F.pad(img.float(), (1, 1, 1, 1), mode="replicate"), self.sobel)
        ix, iy = g[:, 0:1], g[:, 1:2]
        prod = torch.cat([ix * ix, iy * iy, ix * iy], dim=1)
        h = self.bs // 2
        s = F.conv2d(F.pad(prod, (h, h, h, h), mode="replicate"), self.box_sum, groups=3)
        a, c, b = s[:, 0], s[:, 1], s[:, 2]
        half_trace = (a + c) * 0.5
        half_diff = (a - c) * 0.5
        root = torch.sqrt(half_diff * half_diff + b * b + 1e-10)
        return torch.clamp(half_trace - root, min=0.0)                          # (B, H, W)

    # utils/keypoint_utils.py:26-43, :71-115
    def keypoints(self, s: torch.Tensor):
        b, h, w = s.shape
        k = 2 * self.r_nms + 1
        mx = F.max_pool2d(s.unsqueeze(1), kernel_size=k, stride=1, padding=self.r_nms).squeeze(1)
        m = s * (s >= mx - 1e-7).float()
        if self.margin > 0:
            ys = torch.arange(h).view(1, h, 1)
            xs = torch.arange(w).view(1, 1, w)
            keep = (ys >= self.margin) & (ys < h - self.margin) & (xs >= self.margin) & (xs < w - self.margin)
            m = m * keep.float()
        m = torch.where(m > self.thr_score, m, torch.zeros_like(m))
        val, idx = torch.topk(m.reshape(b, -1), self.k, dim=1)
        kp = torch.stack([(idx // w).float(), (idx % w).float()], dim=-1)
        ok = val > 0
        kp = torch.where(ok.unsqueeze(-1), kp, torch.full_like(kp, -1.0))
        return kp, torch.where(ok, val, torch.zeros_like(val))

    # descriptor/bad.py:458-574, non-oriented, sampling_mode="nearest"
    def describe(self, img: torch.Tensor, kp: torch.Tensor) -> torch.Tensor:
        _, _, h, w = img.shape
        valid = (kp[:, :, 0] >= 0).float()
        ky = kp[:, :, 0].clamp(0.0, float(h - 1)).unsqueeze(-1)
        kx = kp[:, :, 1].clamp(0.0, float(w - 1)).unsqueeze(-1)
        sy, sx = 2.0 / (h - 1 + 1e-8), 2.0 / (w - 1 + 1e-8)
        bank = F.conv2d(F.pad(img, (self.rmax,) * 4, mode="replicate"), self.bank)
        x1, x2, y1, y2 = self.off
        sample = []
        for ox, oy in ((x1, y1), (x2, y2)):
            grid = torch.stack([(kx + ox) * sx - 1.0, (ky + oy) * sy - 1.0], dim=-1)
            got = F.grid_sample(bank, grid, mode="nearest", padding_mode="border", align_corners=True)
            sample.append((got * self.sel).sum(dim=1))
        centered = sample[0] - sample[1] - self.thr
        if not self.binarize:
            d = centered
        elif self.soft:
            d = torch.sigmoid(-centered * self.temperature)
        else:
            d = (centered <= 0).float()
        d = d * valid.unsqueeze(-1)
        return F.normalize(d, p=2, dim=-1) if self.normalize else d

    # matching/sinkhorn.py:95-103, :170-206
    def sinkhorn(self, d1: torch.Tensor, d2: torch.Tensor) -> torch.Tensor:
        b, n, _ = d1.shape
        m = d2.shape[1]
        cost = (d1 * d1).sum(-1, keepdim=True) + (d2 * d2).sum(-1).unsqueeze(1) - 2.0 * torch.bmm(d1, d2.transpose(1, 2))
        z = F.pad(-cost.clamp(min=0.0) / self.eps, (0, 1, 0, 1), value=-self.unused / self.eps)
        log_mu = torch.zeros(b, n + 1)
        log_nu = torch.zeros(b, m + 1)
        log_mu[:, n], log_nu[:, m] = math.log(m), math.log(n)
        u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
        for _ in range(self.iters):
            u = log_mu - torch.logsumexp(z + v.unsqueeze(1), dim=2)
            v = log_nu - torch.logsumexp(z + u.unsqueeze(2), dim=1)
        return torch.exp(z + u.unsqueeze(2) + v.unsqueeze(1))

    # matching/match_extraction.py:72-181
    @staticmethod
    def mutual_matches(p: torch.Tensor, k1: torch.Tensor, k2: torch.Tensor, max_matches: int = 100, threshold: float = 0.1):
        core = p[:, :-1, :-1]
        b, n, _ = core.shape
        best_j = core.argmax(dim=2)
        best_i = core.argmax(dim=1)
        rows = torch.arange(n).unsqueeze(0).expand(b, -1)
        mutual = best_i.gather(1, best_j) == rows
        sc = core.gather(2, best_j.unsqueeze(-1)).squeeze(-1)
        sc = torch.where(mutual & (sc >= threshold), sc, torch.zeros_like(sc))
        top, order = torch.topk(sc, min(max_matches, n), dim=1)
        j = best_j.gather(1, order)
        mk1 = k1.gather(1, order.unsqueeze(-1).expand(-1, -1, 2))
        mk2 = k2.gather(1, j.unsqueeze(-1).expand(-1, -1, 2))
        return mk1, mk2, top, top > 0

    @torch.no_grad()
    def forward(self, img1: torch.Tensor, img2: torch.Tensor):
        """(B,1,H,W) float32 x2 -> (keypoints1, keypoints2, P)"""
        k1, _ = self.keypoints(self.scores(img1))
        k2, _ = self.keypoints(self.scores(img2))
        return k1, k2, self.sinkhorn(self.describe(img1, k1), self.describe(img2, k2))

    @torch.no_grad()
    def match(self, img1: torch.Tensor, img2: torch.Tensor, max_matches: int = 100, threshold: float = 0.1):
        k1, k2, p = self.forward(img1, img2)
        return self.mutual_matches(p, k1, k2, max_matches, threshold)


def time_protocol(fn, warmup: int = 5, timed: int = 10) -> float:
    """The reference harness's protocol (sample/image_matching.py:313-328): warm-up runs, then the mean of `timed`
    runs, in seconds per call."""
    import time
    for _ in range(warmup):
        fn()
    t0 = time.perf_counter()
    for _ in range(timed):
        fn()
    return (time.perf_counter() - t0) / timed
