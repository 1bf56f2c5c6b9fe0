"""`pytorch_model` -- the reference's top-level package name, as an alias of onnx_image_processing_amd.pytorch_model.

With the repository root on sys.path, the reference's own import lines work unchanged and resolve to the
MI355X implementation (the very same module objects, not copies):

    from pytorch_model.detector import ShiTomasiScore
    from pytorch_model.descriptor.bad import SparseBAD
    from pytorch_model.matching.outlier_filters import probability_ratio_filter
    from pytorch_model.feature_detection.shi_tomasi_sparse_bad_sinkhorn import ShiTomasiSparseBADSinkhornMatcher

Sub-packages outside the hot path (`vo`, `depth`, ...) do not exist and raise ImportError.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_IMPL = "onnx_image_processing_amd.pytorch_model"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if not name.startswith(__name__ + "."):
            return None
        real = _IMPL + name[len(__name__):]
        try:
            importlib.import_module(real)
        except ImportError:
            return None
        return importlib.util.spec_from_loader(name, self, is_package=hasattr(sys.modules[real], "__path__"))

    def create_module(self, spec):
        return sys.modules[_IMPL + spec.name[len(__name__):]]

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_impl = importlib.import_module(_IMPL)
__doc__ = (__doc__ or "") + "\n" + (_impl.__doc__ or "")
