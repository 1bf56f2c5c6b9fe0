import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_perf: a RATE comparison on a real MI355X (run with -m gpu_perf on a quiet box; "
                                       "not part of -m gpu: a busy box must not fail a correctness suite)")
    # the library is a build product (git-ignored): compile it once if this checkout does not have it yet
    from onnx_image_processing_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        from onnx_image_processing_amd.build import build
        build(verbose=False)


_GUARD_BYTES = 512
_guards = []          # (weakref to the flat allocation, payload bytes) of every live poisoned GPU buffer


def pytest_sessionstart(session):
    """MI_POISON_EMPTY=1: every torch.empty / empty_like / new_empty on the GPU comes back filled with a poison pattern
    (NaN for floating types, 0x5A bytes otherwise) instead of whatever the allocator recycled -- a result that depends
    on the previous content of an output or workspace buffer (a workspace assumed zero, a row a kernel forgot to write)
    then fails a parity test instead of passing by luck -- and sits between two 512-byte guard bands (0xA5) that the
    `_guard_bands_intact` fixture checks after every test: a kernel that writes outside a buffer it was handed
    fails the test that ran it.  A validation mode for the GPU suite, off by default."""
    if os.environ.get("MI_POISON_EMPTY", "0") != "1":
        return
    import weakref

    import torch

    orig_empty = torch.empty

    def guarded(shape, dtype, device):
        numel = 1
        for d in shape:
            numel *= int(d)
        item = torch.empty((), dtype=dtype).element_size()
        flat = orig_empty((numel * item + 2 * _GUARD_BYTES,), dtype=torch.uint8, device=device)
        flat.fill_(0xA5)                               # guard bands in front of and behind the payload
        flat[_GUARD_BYTES: _GUARD_BYTES + numel * item].fill_(0x5A)
        out = (flat[_GUARD_BYTES: _GUARD_BYTES + numel * item].view(dtype).view(*shape) if numel
               else orig_empty(tuple(shape), dtype=dtype, device=device))
        if numel and out.is_floating_point():
            out.fill_(float("nan"))
        out._mi_guard_base = flat                      # keeps the allocation (and its guard band) alive with the view
        _guards.append((weakref.ref(flat), numel * item))
        return out

    def empty(*size, **kw):
        dev = kw.get("device", None)
        plain = set(kw) <= {"dtype", "device"} and dev is not None and torch.device(dev).type == "cuda"
        if not plain:
            return orig_empty(*size, **kw)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        return guarded(shape, kw.get("dtype", None) or torch.get_default_dtype(), dev)

    def empty_like(t, **kw):
        if kw or not t.is_cuda:
            return torch._C._VariableFunctions.empty_like(t, **kw)
        return guarded(tuple(t.shape), t.dtype, t.device)

    orig_new_empty = torch.Tensor.new_empty

    def new_empty(self, *size, **kw):
        if kw or not self.is_cuda:
            return orig_new_empty(self, *size, **kw)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        return guarded(shape, self.dtype, self.device)

    torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty


@pytest.fixture(autouse=True)
def _guard_bands_intact():
    yield
    if not _guards:
        return
    import torch
    torch.cuda.synchronize()
    live = []
    for ref, nbytes in _guards:
        flat = ref()
        if flat is None:
            continue
        live.append((ref, nbytes))
        assert bool((flat[:_GUARD_BYTES] == 0xA5).all()), f"a kernel wrote in front of a {nbytes}-byte buffer (guard band damaged)"
        assert bool((flat[_GUARD_BYTES + nbytes:] == 0xA5).all()), f"a kernel wrote past the end of a {nbytes}-byte buffer (guard band damaged)"
    _guards[:] = live


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
