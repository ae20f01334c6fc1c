import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
dev = torch.device("cuda:0"); N = 252; G = ops.IgemmGeom
B, Ho, Ci, Co, s = {"512": (N, 20, 512, 512, 1), "256": (N, 40, 256, 256, 1), "128": (2 * N, 40, 128, 128, 1)}[os.environ.get("FP_LAYER", "512")]
x = (torch.randn((B, Ho + 2, Ho + 2, Ci), device=dev) * 0.5).half(); w = (torch.randn((Co, 9 * Ci), device=dev) * 0.02).half()
b = torch.randn(Co, device=dev); y = torch.zeros((B, Ho + 2, Ho + 2, Co), dtype=torch.float16, device=dev)
gin = G.image(Ho, Ho, 1, Ci, offset=0); gout = G.image(Ho, Ho, 1, Co)
f = lambda: ops.igemm_f16(x, gin, w, b, y, gout, B * Ho * Ho, Co, Ci, 9, relu=True)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{ms:.3f} ms  ({2.0 * B * Ho * Ho * Co * Ci * 9 / ms / 1e9:.0f} TFLOP/s nominal)")
