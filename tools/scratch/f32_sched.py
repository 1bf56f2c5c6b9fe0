import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from onnx_image_processing_amd import ops
torch.manual_seed(0)
for B in (128, 256, 448):
    d1 = torch.nn.functional.normalize(torch.randn(B, 512, 256, device="cuda"), dim=-1)
    d2 = torch.nn.functional.normalize(torch.randn(B, 512, 256, device="cuda"), dim=-1)
    z, pitch = ops.cost_logscores_f32(d1, d2, 0, 0.05)
    for pin in (2, 0, 1):
        ops.set_sinkhorn_schedule(pin)
        for _ in range(3):
            ops.sinkhorn(z, 512, pitch, -20.0, 20, want_p=False)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.sinkhorn(z, 512, pitch, -20.0, 20, want_p=False)
        e.record(); torch.cuda.synchronize()
        print(B, "pairs, schedule", pin, round(s.elapsed_time(e) / 20, 4), "ms")
    ops.set_sinkhorn_schedule(-1)
