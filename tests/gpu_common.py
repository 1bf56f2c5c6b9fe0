"""Shared by the GPU test files: the device, array upload, the module fixture."""
import os

import numpy as np
import pytest
import torch

from onnx_image_processing_amd.synth import synth_batch, synth_image  # noqa: F401

DEV = "cuda:0"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _status_word(state) -> int:
    """The status word of the mi_sinkhorn_dots call behind `state` (ops.sinkhorn_bits(..., return_state=True)): it lives
    inside the call's workspace; 0 = solved."""
    work, addr = state[4]
    torch.cuda.synchronize()
    return int(work.view(torch.int32)[(addr - work.data_ptr()) // 4].item())


def gpu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def mods():
    from onnx_image_processing_amd import _native
    from onnx_image_processing_amd.pytorch_model.descriptor.bad import SparseBAD
    from onnx_image_processing_amd.pytorch_model.detector import ShiTomasiScore
    from onnx_image_processing_amd.pytorch_model.feature_detection import (
        MatchExtractionWrapper, ShiTomasiSparseBADSinkhornMatcher)
    from onnx_image_processing_amd.pytorch_model.matching import SinkhornMatcher, SinkhornMatcherWithScores
    from onnx_image_processing_amd.pytorch_model.matching.match_extraction import MutualNearestNeighborMatcher
    from onnx_image_processing_amd.pytorch_model.utils import apply_nms_maxpool, select_topk_keypoints
    from onnx_image_processing_amd.pytorch_model.utils.keypoint_utils import detect_keypoints
    _native.load()     # fail loudly if the HIP library is missing
    return dict(locals())


def _images(g):
    a, b = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]), noise=int(g["noise"]))
    if int(g["blank"][0]) >= 0:
        y0, y1, x0, x1 = [int(v) for v in g["blank"]]
        b[:, :, y0:y1, x0:x1] = 77.0
    return a, b
