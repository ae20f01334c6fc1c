import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops, _lib
dev = torch.device("cuda:0")
B = 504
x = (torch.rand((B, 6, 160, 160), device=dev) - 0.5).half()
w = (torch.randn((64, 294), device=dev) * 0.05).half()
b = torch.zeros(64, device=dev); sc = torch.ones(64, device=dev); sh = torch.zeros(64, device=dev)
y = torch.zeros((B, 82, 82, 64), dtype=torch.float16, device=dev)
import ctypes
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "foundationpose_amd", "csrc", "libfp_amd_profile.so"))
for _ in range(3): ops.conv7x7s2_bn_relu(x, w, b, sc, sh, y, 1)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
L.fp_dbg_conv1(out, 1)
ops.conv7x7s2_bn_relu(x, w, b, sc, sh, y, 1)
torch.cuda.synchronize()
L.fp_dbg_conv1(out, 0)
n = out[6]
names = ["stage issue", "wait dma+barrier", "k-loops", "epilogues", "end barrier", "total"]
for i, nm in enumerate(names): print(f"{nm:18s} {out[i] / n:10.0f} cycles per wave ({100.0 * out[i] / out[5]:.1f} %)")
print("waves", n)
