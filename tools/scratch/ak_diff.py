import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from onnx_image_processing_amd import ops, _native as N
from onnx_image_processing_amd.synth import synth_image
for (n, h, w) in ((1, 96, 128), (1, 40, 256), (2, 131, 258)):
    img = np.stack([synth_image(3500 + i, h, w) for i in range(n)])[:, None].astype(np.float32)
    x = torch.from_numpy(img).cuda()
    for iters, nms in ((1, 3), (3, 5)):
        want_l = ops.akaze_diffuse(x, iters, 0.05, 0.25)
        want_s = ops.akaze_hessian_scores(want_l, 0.001, nms)
        got_l, got_s = ops.akaze_scale(x, iters, 0.05, 0.25, 0.001, nms)
        for name, g, wv in (("L", got_l, want_l), ("S", got_s, want_s)):
            bad = (g != wv).nonzero().cpu().numpy()
            print((n, h, w), (iters, nms), name, "mismatches", len(bad), "of", g.numel())
            if len(bad):
                ys, xs = bad[:, 2], bad[:, 3]
                print("   rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "first", bad[:5].tolist())
                print("   col histogram mod 2:", np.bincount(xs % 2, minlength=2), " distinct cols:", len(set(xs.tolist())), "distinct rows", len(set(ys.tolist())))
                i = bad[0]
                print("   got", g[tuple(i)].item(), "want", wv[tuple(i)].item())
# DPP direction probe through a tiny torch-free kernel is not available here; infer from the mismatch pattern
