"""The two row-owning fused kernels of the refiner's encoder layer (csrc/linear_ln.hip) against the launches they replace, at the
refiner's shapes (M = 126 x 400 and 252 x 400 rows):
  fp_linear_layernorm_fwd   : out_proj + residual + norm1                               (fp_igemm_f16_fwd + fp_layernorm_res_fwd)
  fp_ffn_layernorm_mean_fwd : linear1 + ReLU + linear2 + residual + norm2 + token mean  (2 x fp_igemm_f16_fwd + fp_colmean_f16_fwd)
    python scripts/bench_linear_ln.py
The first is bit-identical to its two-kernel path, the second equal up to the fp32 summation order of the token mean
(tests/test_gpu_parity.py); checked again here.  Per-phase times of a tile: scripts/dbg_linear_ln.py (profiling build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
from foundationpose_amd.engine import _HipLinear

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
mk = lambda: _HipLinear((torch.randn((512, 512), generator=g) * 0.05).to(dev), (torch.randn((512,), generator=g) * 0.1).to(dev))
lin, l1, l2 = mk(), mk(), mk()
pk, p1, p2 = (ops.PackedLinear512(m.w) for m in (lin, l1, l2))
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


print("lib", os.environ.get("FP_AMD_LIB", "libfp_amd.so"))
for n in (126, 252):
    x = torch.randn((n, 400, 512), generator=g).to(torch.float16).to(dev)
    tok = torch.randn((n, 400, 512), generator=g).to(torch.float16).to(dev)
    y32 = torch.randn((n, 400, 512), generator=g).to(dev)
    two = lambda: ops.layernorm_res(lin(x), gamma, beta, 1e-5, tok16=tok, pe=pe)
    one = lambda: ops.linear_layernorm_res(x, pk, lin.b, gamma, beta, 1e-5, tok16=tok, pe=pe)
    a, b = two(), one()
    same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    t_lin, t_two, t_one = timed(lambda: lin(x)), timed(two), timed(one)
    gf = 2.0 * n * 400 * 512 * 512 / 1e9
    print(f"N={n}: linear {t_lin:.1f} us, linear + layernorm {t_two:.1f} us, fused {t_one:.1f} us ({gf / t_one * 1e-3:.0f} TFLOP/s), "
          f"bit-identical: {same}")
    three = lambda: ops.colmean_f16(l2(l1(x, relu=True)), gamma, beta, 1e-5, resid32=y32)
    ffn = lambda: ops.ffn_layernorm_mean(x, p1, l1.b, p2, l2.b, y32, gamma, beta, 1e-5)
    a, b = three().clone(), ffn().clone()
    t3, t1 = timed(three), timed(ffn)
    print(f"       FFN: linear1 + linear2 + colmean {t3:.1f} us, fused + finish {t1:.1f} us ({2 * gf / t1 * 1e-3:.0f} TFLOP/s), "
          f"max |diff| {float((a - b).abs().max()):.3e} (values up to {float(a.abs().max()):.3f})")
