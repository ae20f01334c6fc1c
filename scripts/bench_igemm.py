"""Per-layer throughput of fp_igemm_f16_fwd at the bench shapes (N=252) + the other hand-written kernels.  Debug/profiling aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops

dev = torch.device("cuda:0")
N = int(os.environ.get("FP_N", "252"))
G = ops.IgemmGeom


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


layers = [("c2   64->128 s2 80->40", 2 * N, 40, 64, 128, 2), ("stem 128->128 40", 2 * N, 40, 128, 128, 1),
          ("j    256->256 40", N, 40, 256, 256, 1), ("j2   256->512 s2 40->20", N, 20, 256, 512, 2), ("j    512->512 20", N, 20, 512, 512, 1)]
for name, B, Ho, Ci, Co, s in layers:
    Hi = Ho * s
    x = (torch.randn((B, Hi + 2, Hi + 2, Ci), device=dev) * 0.5).half()
    w = (torch.randn((Co, 9 * Ci), device=dev) * 0.02).half()
    b = torch.randn(Co, device=dev)
    y = torch.zeros((B, Ho + 2, Ho + 2, Co), dtype=torch.float16, device=dev)
    r = (torch.randn((B, Ho + 2, Ho + 2, Co), device=dev) * 0.5).half()
    gin = G.image(Ho, Ho, 1, Ci, stride=s, offset=0); gout = G.image(Ho, Ho, 1, Co)
    M = B * Ho * Ho
    for res in (False, True):
        ms = timeit(lambda: ops.igemm_f16(x, gin, w, b, y, gout, M, Co, Ci, 9, relu=True, residual=r if res else None, r_geom=gout if res else None))
        print(f"igemm conv {name} res={int(res)}: {ms:.3f} ms  {2.0 * M * Co * Ci * 9 / ms / 1e9:.0f} TFLOP/s", flush=True)
for M, K, No in [(N * 400, 512, 1536), (N * 400, 512, 512)]:
    x = torch.randn((M, K), device=dev).half(); w = (torch.randn((No, K), device=dev) * 0.05).half(); b = torch.randn(No, device=dev)
    y = torch.empty((M, No), dtype=torch.float16, device=dev)
    ms = timeit(lambda: ops.igemm_f16(x, G.matrix(K), w, b, y, G.matrix(No), M, No, K, 1))
    print(f"igemm linear M={M} K={K} N={No}: {ms:.3f} ms  {2.0 * M * K * No / ms / 1e9:.0f} TFLOP/s", flush=True)
    ms = timeit(lambda: torch.nn.functional.linear(x, w, b.half()))
    print(f"torch  linear M={M} K={K} N={No}: {ms:.3f} ms  {2.0 * M * K * No / ms / 1e9:.0f} TFLOP/s", flush=True)
if os.environ.get('FP_LAYERS_ONLY'):
    sys.exit(0)
# whole encoder + plan
from foundationpose_amd import engine
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
from foundationpose_amd.refine_network import RefineNet
cfg = dict(DEFAULT_REFINE_CFG)
net = RefineNet(cfg=cfg, c_in=6); net.load_state_dict(random_state_dict("refine", cfg, 0))
plan = engine.RefinePlan(net, dev, precision="fp16")
AB = torch.rand((2 * N, 6, 160, 160), device=dev).half()
print(f"HipEncoder: {timeit(lambda: plan.enc(AB), 5):.3f} ms", flush=True)
print(f"RefinePlan: {timeit(lambda: plan(AB), 5):.3f} ms", flush=True)
with ops.KernelTimers() as t:
    plan(AB)
for k, v in t.summary().items():
    print(f"   {k}: calls {v['calls']} avg {v['avg_ms']:.3f} ms  {v['flops'] / max(v['avg_ms'], 1e-9) / 1e9:.0f} TFLOP/s")
