import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library is a build product (git-ignored): compile it once if this checkout does not have it yet
    from onnx_image_processing_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        from onnx_image_processing_amd.build import build
        build(verbose=False)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
