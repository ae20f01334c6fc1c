"""fp_linear_layernorm_fwd (out_proj + residual + LayerNorm in one launch) against the two launches it replaces, at the
refiner's shapes (M = 126 x 400 and 252 x 400 rows, K = 512):
    python scripts/bench_linear_ln.py                                   # product tile (128 rows, one workgroup per CU)
    make -C foundationpose_amd/csrc profile
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so FP_LL_TILE=64 python scripts/bench_linear_ln.py   # 64 rows, two per CU
Both tiles are bit-identical to the two-kernel path (tests/test_gpu_parity.py::test_linear_layernorm_is_the_two_kernel_path passes
with either; checked again here).  Alone on the chip the numbers say little about the step: the question the 64-row tile answers is
what happens next to the other sub-batch's GEMMs, so compare `python bench.py --steps 10 --no-cpu-baseline --no-kernel-table`
with and without FP_LL_TILE=64 as well."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
from foundationpose_amd.engine import _HipLinear

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
lin = _HipLinear((torch.randn((512, 512), generator=g) * 0.05).to(dev), (torch.randn((512,), generator=g) * 0.1).to(dev))
gamma = (1.0 + 0.1 * torch.randn((512,), generator=g)).to(dev)
beta = (0.1 * torch.randn((512,), generator=g)).to(dev)
pe = torch.randn((400, 512), generator=g).to(dev)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("lib", os.environ.get("FP_AMD_LIB", "libfp_amd.so"), "FP_LL_TILE", os.environ.get("FP_LL_TILE", "-"))
for n in (126, 252):
    x = torch.randn((n, 400, 512), generator=g).to(torch.float16).to(dev)
    tok = torch.randn((n, 400, 512), generator=g).to(torch.float16).to(dev)
    two = lambda: ops.layernorm_res(lin(x), gamma, beta, 1e-5, tok16=tok, pe=pe)
    one = lambda: ops.linear_layernorm_res(x, lin.w, lin.b, gamma, beta, 1e-5, tok16=tok, pe=pe)
    a, b = two(), one()
    same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    t_lin, t_two, t_one = timed(lambda: lin(x)), timed(two), timed(one)
    gf = 2.0 * n * 400 * 512 * 512 / 1e9
    print(f"N={n}: linear {t_lin:.1f} us, linear + layernorm {t_two:.1f} us, fused {t_one:.1f} us ({gf / t_one * 1e-3:.0f} TFLOP/s), "
          f"bit-identical: {same}")
