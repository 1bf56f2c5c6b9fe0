"""SparseBAD -- mirror of reference pytorch_model/descriptor/bad.py:336-576."""
import torch
from torch import nn

from ... import _native as N
from ... import ops
from .bad_params import _get_bad_learned_params


class SparseBAD(nn.Module):
    """BAD (Box Average Difference) descriptors at keypoints only.

    forward(image (B,1,H,W), keypoints (B,K,2) as (y,x), orientation=None) -> (B,K,num_pairs)
    float32.  Constructor arguments, validation and buffer names follow bad.py:374-434.  The
    K4 kernel evaluates every box from an exact summed-area table of a 34x34 window around the
    keypoint instead of the reference's dense 8-channel box-mean bank, so `box_kernel_bank` and
    `radius_select` exist only for state_dict() compatibility.
    """

    def __init__(
        self,
        num_pairs: int = 256,
        binarize: bool = False,
        soft_binarize: bool = True,
        temperature: float = 10.0,
        normalize_descriptors: bool = True,
        sampling_mode: str = "nearest",
    ):
        super().__init__()
        if num_pairs not in (256, 512):
            raise ValueError(f"num_pairs must be 256 or 512 to use learned BAD patterns, got {num_pairs}")
        if sampling_mode not in ("nearest", "bilinear"):
            raise ValueError(f"sampling_mode must be 'nearest' or 'bilinear', got {sampling_mode}")
        self.num_pairs = num_pairs
        self.binarize = binarize
        self.soft_binarize = soft_binarize
        self.temperature = temperature
        self.normalize_descriptors = normalize_descriptors
        self.sampling_mode = sampling_mode

        box, thr = _get_bad_learned_params(num_pairs)
        for name, col in (("offset_x1", 0), ("offset_x2", 1), ("offset_y1", 2), ("offset_y2", 3)):
            self.register_buffer(name, box[:, col] - 16.0)
        self.register_buffer("radii", box[:, 4].to(torch.int64))
        self.register_buffer("thresholds", thr)
        for name in ("offset_y1", "offset_x1", "offset_y2", "offset_x2", "thresholds"):
            self.register_buffer(name + "_v", getattr(self, name).view(1, 1, -1))
        self.max_radius = int(self.radii.max().item())
        sel = torch.zeros(self.max_radius + 1, num_pairs)
        sel[self.radii, torch.arange(num_pairs)] = 1.0
        self.register_buffer("radius_select", sel)
        r = torch.arange(self.max_radius + 1, dtype=torch.float32).view(-1, 1, 1)
        c = torch.arange(-self.max_radius, self.max_radius + 1, dtype=torch.float32)
        inside = ((c.abs().view(1, -1, 1) <= r) & (c.abs().view(1, 1, -1) <= r)).float()
        self.register_buffer("box_kernel_bank", (inside / (2.0 * r + 1.0) ** 2).unsqueeze(1))
        # packed geometry consumed by mi_sparse_bad: x1 | x2<<5 | y1<<10 | y2<<15 | r<<20
        b = box.to(torch.int64)
        geom = b[:, 0] | (b[:, 1] << 5) | (b[:, 2] << 10) | (b[:, 3] << 15) | (b[:, 4] << 20)
        self.register_buffer("pair_geom", geom.to(torch.int32), persistent=False)
        self.register_buffer("pair_thr", thr.clone(), persistent=False)
        # largest |offset from the patch centre| + box radius of the table, in pixels (rounded up a hair): the bound
        # mi_sparse_bad_oriented uses to size its per-keypoint window (22.22 for both learned tables)
        off = box[:, :4] - 16.0
        reach = torch.maximum(torch.hypot(off[:, 0], off[:, 2]), torch.hypot(off[:, 1], off[:, 3])) + box[:, 4]
        self.max_reach = float(reach.max().item()) + 1e-3
        self._plan = None          # device-side fast-path plan, built lazily per device
        self.use_fast_path = True  # tests switch this off to exercise the general kernel path

    @property
    def mode(self) -> int:
        if not self.binarize:
            return N.MI_BAD_RAW
        return N.MI_BAD_SOFT if self.soft_binarize else N.MI_BAD_HARD

    def _check(self, image: torch.Tensor, orientation):
        if self.pair_geom.device != image.device:
            raise RuntimeError(
                f"SparseBAD buffers are on {self.pair_geom.device} but the image is on {image.device}; "
                "move the module with .to(device)"
            )

    def _get_plan(self):
        if not self.use_fast_path or self.mode != N.MI_BAD_HARD:
            return None
        if self._plan is None or self._plan.device != self.pair_geom.device:
            self._plan = ops.bad_plan(self.pair_geom, self.pair_thr)
        return self._plan

    @torch.no_grad()
    def forward(self, image: torch.Tensor, keypoints: torch.Tensor, orientation: torch.Tensor | None = None):
        """orientation: None (non-oriented), the reference's dense angle map (B,1,H,W) in radians, or --
        an extension -- the angles at the keypoints themselves (B,K)."""
        self._check(image, orientation)
        bilinear = self.sampling_mode == "bilinear"
        if orientation is None and bilinear:             # bilinear, non-oriented: angle 0 leaves the offsets as they are
            orientation = torch.zeros(keypoints.shape[:2], dtype=torch.float32, device=image.device)
        if orientation is not None:                      # oriented branch, bad.py:487-517
            desc, _ = ops.sparse_bad_oriented(image, keypoints, orientation, self.pair_geom, self.pair_thr, self.mode,
                                              self.temperature, self.normalize_descriptors, bilinear=bilinear,
                                              max_reach=self.max_reach)
            return desc
        desc, _ = ops.sparse_bad(image, keypoints, self.pair_geom, self.pair_thr, self.mode, self.temperature,
                                 self.normalize_descriptors, want_desc=True, want_bits=False, plan=self._get_plan())
        return desc

    @torch.no_grad()
    def forward_bits(self, image: torch.Tensor, keypoints: torch.Tensor,
                     orientation: torch.Tensor | None = None) -> torch.Tensor:
        """Hard-binarised descriptors as packed bits (B,K,num_pairs/32) int32 -- the form the
        bit-exact cost kernel consumes.  Only meaningful for binarize=True, soft_binarize=False."""
        if self.mode != N.MI_BAD_HARD:
            raise RuntimeError("forward_bits needs binarize=True, soft_binarize=False")
        self._check(image, orientation)
        bilinear = self.sampling_mode == "bilinear"
        if orientation is None and bilinear:
            orientation = torch.zeros(keypoints.shape[:2], dtype=torch.float32, device=image.device)
        if orientation is not None:
            _, bits = ops.sparse_bad_oriented(image, keypoints, orientation, self.pair_geom, self.pair_thr,
                                              N.MI_BAD_HARD, self.temperature, self.normalize_descriptors,
                                              want_desc=False, want_bits=True, bilinear=bilinear,
                                              max_reach=self.max_reach)
            return bits
        _, bits = ops.sparse_bad(image, keypoints, self.pair_geom, self.pair_thr, N.MI_BAD_HARD, self.temperature,
                                 self.normalize_descriptors, want_desc=False, want_bits=True, plan=self._get_plan())
        return bits


class BADDescriptor(nn.Module):
    """Dense BAD descriptor map -- mirror of reference descriptor/bad.py:14-218.

    forward(x (B,1,H,W), orientation=None) -> (B, num_pairs, H, W): the centred BAD response
    at every pixel (raw / sigmoid(-c*T) / (c <= 0); no normalisation).  Buffer names follow
    bad.py:31-60.  Box sums are exact (fp64 summed-area table per tile); the reference's fp32
    integral image is off by up to ~1 intensity unit on large images, so parity with it is by
    tolerance / bit-agreement rate.  With an orientation map every pixel's offsets are rotated and the
    box means sampled bilinearly (bad.py:112-187).
    """

    def __init__(self, num_pairs: int = 256, binarize: bool = False, soft_binarize: bool = True,
                 temperature: float = 10.0) -> None:
        super().__init__()
        self.num_pairs = num_pairs
        self.binarize = binarize
        self.soft_binarize = soft_binarize
        self.temperature = temperature
        box, thr = _get_bad_learned_params(num_pairs)
        for name, col in (("offset_x1", 0), ("offset_x2", 1), ("offset_y1", 2), ("offset_y2", 3)):
            self.register_buffer(name, box[:, col] - 16.0)
        self.register_buffer("radii", box[:, 4].to(torch.int64))
        self.register_buffer("thresholds", thr)
        self.register_buffer("area", ((2.0 * self.radii.float() + 1.0) ** 2).view(-1, 1, 1))
        self.max_radius = int(self.radii.max().item())
        sel = torch.zeros(self.max_radius + 1, num_pairs)
        sel[self.radii, torch.arange(num_pairs)] = 1.0
        self.register_buffer("radius_select", sel)
        r = torch.arange(self.max_radius + 1, dtype=torch.float32).view(-1, 1, 1)
        c = torch.arange(-self.max_radius, self.max_radius + 1, dtype=torch.float32)
        inside = ((c.abs().view(1, -1, 1) <= r) & (c.abs().view(1, 1, -1) <= r)).float()
        self.register_buffer("box_kernel_bank", (inside / (2.0 * r + 1.0) ** 2).unsqueeze(1))
        b = box.to(torch.int64)
        geom = b[:, 0] | (b[:, 1] << 5) | (b[:, 2] << 10) | (b[:, 3] << 15) | (b[:, 4] << 20)
        self.register_buffer("pair_geom", geom.to(torch.int32), persistent=False)
        self.register_buffer("pair_thr", thr.clone(), persistent=False)

    @property
    def mode(self) -> int:
        if not self.binarize:
            return N.MI_BAD_RAW
        return N.MI_BAD_SOFT if self.soft_binarize else N.MI_BAD_HARD

    @torch.no_grad()
    def forward(self, x: torch.Tensor, orientation: torch.Tensor | None = None) -> torch.Tensor:
        if orientation is not None:                      # per-pixel rotation + bilinear sampling, bad.py:112-187
            return ops.bad_dense_oriented(x, orientation, self.pair_geom, self.pair_thr, self.mode, self.temperature)
        return ops.bad_dense(x, self.pair_geom, self.pair_thr, self.mode, self.temperature)

    @torch.no_grad()
    def at_keypoints(self, x: torch.Tensor, keypoints: torch.Tensor) -> torch.Tensor:
        """forward(x) sampled at integer keypoints, without building the (B,P,H,W) map: (B,K,P).
        Invalid keypoints (-1,-1) give zero rows (the matcher's masking, shi_tomasi_bad_sinkhorn.py:147)."""
        desc, _ = ops.sparse_bad(x, keypoints, self.pair_geom, self.pair_thr, self.mode, self.temperature, False)
        return desc


@torch.no_grad()
def extract_descriptors_at_keypoints(descriptor_map: torch.Tensor, keypoints: torch.Tensor) -> torch.Tensor:
    """(B,D,H,W), integer keypoints (B,N,2) as (y,x) -> (B,N,D) (bad.py:221-274)."""
    return ops.gather_descriptors(descriptor_map, keypoints, bilinear=False)


@torch.no_grad()
def extract_descriptors_at_keypoints_subpixel(descriptor_map: torch.Tensor, keypoints: torch.Tensor) -> torch.Tensor:
    """(B,D,H,W), float keypoints (B,N,2) -> (B,N,D) by bilinear grid_sample (bad.py:277-333)."""
    return ops.gather_descriptors(descriptor_map, keypoints, bilinear=True)
