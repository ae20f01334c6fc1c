"""phase timers of k_conv_sw (profiling build): cold start / main loop / epilogue per workgroup, and the epilogue's own phases, 100 MHz wall clock
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so python scripts/dbg_conv_sw.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
dev = torch.device("cuda:0")
L = C.CDLL(os.environ["FP_AMD_LIB"])
G = ops.IgemmGeom.image
HAVE_EPI = hasattr(L, "fp_dbg_conv_sw_epilogue")
for (B, H, Ci, Co) in ((504, 40, 128, 128), (252, 40, 256, 256), (252, 20, 512, 512), (126, 40, 256, 256), (126, 20, 512, 512)):
    x = torch.relu(torch.randn((B, H + 2, H + 2, Ci), device=dev) * 0.5).half()
    x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
    w = (torch.randn((Co, 9 * Ci), device=dev) * 0.02).half()
    b = torch.zeros(Co, device=dev)
    r = (torch.randn((B, H + 2, H + 2, Co), device=dev) * 0.5).half()
    y = torch.zeros((B, H + 2, H + 2, Co), dtype=torch.float16, device=dev)
    M = B * H * H
    for res in (True, False):
        def run():
            ops.igemm_f16(x, G(H, H, 1, Ci, stride=1, offset=0), w, b, y, G(H, H, 1, Co), M, Co, Ci, 9, relu=True,
                          residual=r if res else None, r_geom=G(H, H, 1, Co) if res else None, conv_rounding=True)
        for _ in range(3): run()
        torch.cuda.synchronize()
        out, epi = (C.c_ulonglong * 8)(), (C.c_ulonglong * 8)()
        L.fp_dbg_conv_sw(out, 1)
        if HAVE_EPI: L.fp_dbg_conv_sw_epilogue(epi, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        L.fp_dbg_conv_sw(out, 0)
        if HAVE_EPI: L.fp_dbg_conv_sw_epilogue(epi, 0)
        n = out[3]
        line = (f"B={B} {Ci}->{Co} {'+res' if res else 'nores'}: {e0.elapsed_time(e1) * 1e3:.1f} us; {n} tiles = {n / 256:.2f} rounds; per tile: cold {out[0] / n * 10:.0f} ns, loop {out[1] / n * 10:.0f} ns, "
                f"epilogue+drain {out[2] / n * 10:.0f} ns; span {(out[7] - out[6]) * 10 / 1e3:.1f} us")
        if HAVE_EPI and epi[4]:
            m = epi[4]
            line += (f"; epilogue: row tables {epi[0] / m * 10:.0f}, residual requests {epi[1] / m * 10:.0f}, acc->E {epi[2] / m * 10:.0f}, "
                     f"E->stores issued {epi[3] / m * 10:.0f}, drain {(out[2] - epi[0] - epi[1] - epi[2] - epi[3]) / m * 10:.0f} ns")
        print(line, flush=True)
