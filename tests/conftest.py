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


def pytest_sessionstart(session):
    """MI_POISON_EMPTY=1: every torch.empty / empty_like / new_empty on the GPU comes back filled with a poison pattern
    (NaN for floating types, 0x5A bytes otherwise) instead of whatever the allocator recycled -- a result that depends
    on the previous content of an output or workspace buffer (a workspace assumed zero, a row a kernel forgot to write)
    then fails a parity test instead of passing by luck.  A validation mode for the GPU suite, off by default."""
    if os.environ.get("MI_POISON_EMPTY", "0") != "1":
        return
    import torch

    def poison(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            else:
                t.view(torch.uint8).fill_(0x5A)
        return t

    for name in ("empty", "empty_like"):
        orig = getattr(torch, name)
        setattr(torch, name, (lambda o: lambda *a, **k: poison(o(*a, **k)))(orig))
    orig_new = torch.Tensor.new_empty
    torch.Tensor.new_empty = lambda self, *a, **k: poison(orig_new(self, *a, **k))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
