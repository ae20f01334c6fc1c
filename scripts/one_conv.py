"""Runs one igemm conv layer a few times (target of rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
dev = torch.device("cuda:0")
N = 252
G = ops.IgemmGeom
which = os.environ.get("FP_LAYER", "512")
B, Ho, Ci, Co, s = {"512": (N, 20, 512, 512, 1), "256": (N, 40, 256, 256, 1), "128": (2 * N, 40, 128, 128, 1)}[which]
Hi = Ho * s
x = torch.relu(torch.randn((B, Hi + 2, Hi + 2, Ci), device=dev) * 0.5).half()
x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
w = (torch.randn((Co, 9 * Ci), device=dev) * 0.02).half()
b = torch.randn(Co, device=dev)
y = torch.zeros((B, Ho + 2, Ho + 2, Co), dtype=torch.float16, device=dev)
gin = G.image(Ho, Ho, 1, Ci, stride=s, offset=0); gout = G.image(Ho, Ho, 1, Co)
for _ in range(4):
    ops.igemm_f16(x, gin, w, b, y, gout, B * Ho * Ho, Co, Ci, 9, relu=True, conv_rounding=True)
torch.cuda.synchronize()
print("ok")
