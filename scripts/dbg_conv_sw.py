"""phase timers of k_conv_sw (profiling build): cold start / main loop / epilogue per workgroup, 100 MHz wall clock
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so python scripts/dbg_conv_sw.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
dev = torch.device("cuda:0")
L = C.CDLL(os.environ["FP_AMD_LIB"])
G = ops.IgemmGeom.image
for (B, H, Ci, Co) in ((504, 40, 128, 128), (252, 40, 256, 256), (252, 20, 512, 512), (126, 40, 256, 256), (126, 20, 512, 512)):
    x = torch.relu(torch.randn((B, H + 2, H + 2, Ci), device=dev) * 0.5).half()
    x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
    w = (torch.randn((Co, 9 * Ci), device=dev) * 0.02).half()
    b = torch.zeros(Co, device=dev)
    r = (torch.randn((B, H + 2, H + 2, Co), device=dev) * 0.5).half()
    y = torch.zeros((B, H + 2, H + 2, Co), dtype=torch.float16, device=dev)
    M = B * H * H
    run = lambda: ops.igemm_f16(x, G(H, H, 1, Ci, stride=1, offset=0), w, b, y, G(H, H, 1, Co), M, Co, Ci, 9, relu=True, residual=r, r_geom=G(H, H, 1, Co), conv_rounding=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    L.fp_dbg_conv_sw(out, 1)
    run(); torch.cuda.synchronize()
    L.fp_dbg_conv_sw(out, 0)
    n = out[3]
    print(f"B={B} {Ci}->{Co}: {n} tiles = {n / 256:.2f} rounds; per tile: cold start {out[0] / n * 10:.0f} ns, main loop {out[1] / n * 10:.0f} ns, "
          f"epilogue+drain {out[2] / n * 10:.0f} ns; kernel span {(out[7] - out[6]) * 10 / 1e3:.1f} us; sum per tile x rounds {(out[0] + out[1] + out[2]) / n * 10 / 1e3 * -(-n // 256):.1f} us", flush=True)
