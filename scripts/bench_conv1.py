"""patch-embed conv timing (HIP events, 20 launches): images = 2 x hypotheses"""
import hashlib, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
dev = torch.device("cuda:0")
for B in (252, 504):
    torch.manual_seed(B)
    x = (torch.rand((B, 6, 160, 160), device=dev) - 0.5).half()
    w = (torch.randn((64, 294), device=dev) * 0.05).half()
    b = torch.zeros(64, device=dev); sc = torch.ones(64, device=dev); sh = torch.zeros(64, device=dev)
    y = torch.zeros((B, 82, 82, 64), dtype=torch.float16, device=dev)
    for _ in range(3):
        ops.conv7x7s2_bn_relu(x, w, b, sc, sh, y, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv7x7s2_bn_relu(x, w, b, sc, sh, y, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byt = B * (6 * 160 * 160 + 64 * 80 * 80) * 2
    dg = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    print(json.dumps({"lib": os.path.basename(os.environ.get("FP_AMD_LIB", "libfp_amd.so")), "digest": dg, "images": B, "ms": round(ms, 4), "GBps": round(byt / ms / 1e6, 1), "frac_hbm": round(byt / ms / 1e6 / 8000, 3), "TFLOPs": round(2.0 * B * 6400 * 64 * 294 / ms / 1e9, 1)}))
