"""Per-layer throughput of fp_igemm_f16_fwd at the bench shapes (N=252), HIP events on the launch stream.  Profiling aid.
    python scripts/bench_igemm.py                                   # the product library (shifted-window 3x3 kernel)
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so FP_IGEMM_TILE=generic python scripts/bench_igemm.py
                                                                    # profiling build: the generic implicit GEMM everywhere
Prints one JSON object per layer; activations are post-ReLU-like (half zeros), weights random."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops

dev = torch.device("cuda:0")
N = int(os.environ.get("FP_N", "252"))
G = ops.IgemmGeom
tag = os.environ.get("FP_IGEMM_TILE", "default") + "/" + os.path.basename(os.environ.get("FP_AMD_LIB", "libfp_amd.so")) + \
    (":" + os.environ["FP_CONV_SW"] if "FP_CONV_SW" in os.environ else "") + f"/N={N}"


def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


layers = [("c2 64->128 s2", 2 * N, 40, 64, 128, 2), ("stem 128->128", 2 * N, 40, 128, 128, 1), ("joint 256->256", N, 40, 256, 256, 1),
          ("j2 256->512 s2", N, 20, 256, 512, 2), ("joint 512->512", N, 20, 512, 512, 1)]
tot_ms = tot_fl = 0.0
counts = {"c2 64->128 s2": 1, "stem 128->128": 4, "joint 256->256": 4, "j2 256->512 s2": 1, "joint 512->512": 4}
for name, B, Ho, Ci, Co, s in layers:
    Hi = Ho * s
    x = torch.relu(torch.randn((B, Hi + 2, Hi + 2, Ci), device=dev) * 0.5).half()
    x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
    w = (torch.randn((Co, 9 * Ci), device=dev) * 0.02).half()
    b = torch.randn(Co, device=dev).half().float()
    sc, sh = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev) * 0.1
    y = torch.zeros((B, Ho + 2, Ho + 2, Co), dtype=torch.float16, device=dev)
    r = (torch.randn((B, Ho + 2, Ho + 2, Co), device=dev) * 0.5).half()
    gin = G.image(Ho, Ho, 1, Ci, stride=s, offset=0); gout = G.image(Ho, Ho, 1, Co)
    M = B * Ho * Ho
    wt = ops.pack_conv3x3_tiles(w, Co, Ci) if (os.environ.get("FP_W_TILES") == "1" and s == 1) else None    # FP_W_TILES=1: tile-packed weights
    for res in (False, True):
        ms = timeit(lambda: ops.igemm_f16(x, gin, w, b, y, gout, M, Co, Ci, 9, relu=True, residual=r if res else None,
                                          r_geom=gout if res else None, bn_scale=sc, bn_shift=sh, conv_rounding=True, w_tiles=wt))
        fl = 2.0 * M * Co * Ci * 9
        print(json.dumps(dict(lib=tag, layer=name, residual=res, ms=round(ms, 4), TFLOPs=round(fl / ms / 1e9, 1))), flush=True)
        if res == (name.startswith("stem") or name.startswith("joint")):   # roughly the mix of the encoder (half the block convs add the identity)
            pass
    tot_ms += counts[name] * ms; tot_fl += counts[name] * fl
print(json.dumps(dict(lib=tag, layer="encoder 3x3 convs (weighted, residual variant)", ms=round(tot_ms, 3), TFLOPs=round(tot_fl / tot_ms / 1e9, 1))), flush=True)
for M, K, No in [(N * 400, 512, 1536), (N * 400, 512, 512)]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((No, K), device=dev) * 0.05).half(); b = torch.randn(No, device=dev).half().float()
    y = torch.empty((M, No), dtype=torch.float16, device=dev)
    ms = timeit(lambda: ops.igemm_f16(x, G.matrix(K), w, b, y, G.matrix(No), M, No, K, 1))
    print(json.dumps(dict(lib=tag, layer=f"linear M={M} K={K} N={No}", ms=round(ms, 4), TFLOPs=round(2.0 * M * K * No / ms / 1e9, 1))), flush=True)
