import struct, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from onnx_image_processing_amd import _native as N
def bits(x): return struct.unpack("<I", struct.pack("<f", x))[0]
lib = N.use_debug_library()
for which in (0, 4, 5):
    bad = torch.zeros(1, dtype=torch.int64, device="cuda"); first = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    N.check(lib.mi_debug_akaze_math_check(which, 0.05, bits(1e-8), bits(2.0 ** 24), bad.data_ptr(), first.data_ptr(), N.stream_ptr()), "x")
    print(which, int(bad.item()), hex(int(first.item()) & 0xFFFFFFFF))
