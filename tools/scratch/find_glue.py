import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from onnx_image_processing_amd import distributed as D
from onnx_image_processing_amd.pytorch_model.feature_detection import MatchExtractionWrapper, ShiTomasiSparseBADSinkhornMatcher
from onnx_image_processing_amd.synth import synth_batch
B = 448
a, b = synth_batch(1000, B, 480, 640)
i1, i2 = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
model = MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher(max_keypoints=512, **bench.CFG), max_matches=100, match_threshold=0.1).cuda()
for _ in range(14):
    D.pack_records(*model(i1, i2))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    D.pack_records(*model(i1, i2))
    torch.cuda.synchronize()
for e in prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60).splitlines():
    print(e[:260])
