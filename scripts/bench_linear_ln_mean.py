"""The feed-forward half of the refiner's encoder layer on the fused 128 x 512 tile (profiling build only, csrc/linear_ln.hip):
  fp_linear_layernorm_mean_fwd : linear2 + residual + norm2 + token mean            (product: fp_igemm_f16_fwd + fp_colmean_f16_fwd)
  fp_ffn_layernorm_mean_fwd    : linear1 + ReLU + linear2 + residual + norm2 + mean (product: 2 x fp_igemm_f16_fwd + fp_colmean_f16_fwd)
each one launch + a finish kernel:
    make -C foundationpose_amd/csrc profile
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so python scripts/bench_linear_ln_mean.py
Prints the largest deviation (the token mean is summed in another, fixed, order: per-tile partial sums, so the results are not
bit-identical) and the times.  Not wired into the plans: it has to pass the parity gates of tests/test_gpu_amp.py first."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import _lib, ops
from foundationpose_amd.engine import _HipLinear

dev = torch.device("cuda:0")
L = _lib.lib()
fn = L.fp_linear_layernorm_mean_fwd          # AttributeError with the product library: it does not export this
fn.restype = C.c_int
vp = C.c_void_p
fn.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, vp]
ffn = L.fp_ffn_layernorm_mean_fwd
ffn.restype = C.c_int
ffn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]
g = torch.Generator(device="cpu").manual_seed(4)
lin = _HipLinear((torch.randn((512, 512), generator=g) * 0.05).to(dev), (torch.randn((512,), generator=g) * 0.1).to(dev))
lin1 = _HipLinear((torch.randn((512, 512), generator=g) * 0.05).to(dev), (torch.randn((512,), generator=g) * 0.1).to(dev))
gamma = (1.0 + 0.1 * torch.randn((512,), generator=g)).to(dev)
beta = (0.1 * torch.randn((512,), generator=g)).to(dev)


def timed(f, reps=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n, S in ((126, 400), (252, 400), (3, 400), (2, 144)):
    x = torch.randn((n, S, 512), generator=g).to(torch.float16).to(dev)
    y32 = torch.randn((n, S, 512), generator=g).to(dev)
    out = torch.empty((n, 512), dtype=torch.float32, device=dev)
    ws = torch.empty((n * S // 16, 512), dtype=torch.float32, device=dev)

    def fused():
        st = fn(x.data_ptr(), lin.w.data_ptr(), lin.b.data_ptr(), y32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, out.data_ptr(),
                ws.data_ptr(), ws.numel() * 4, n, S, 512, 512, torch.cuda.current_stream().cuda_stream)
        assert st == 0, L.fp_last_error()
        return out

    two = lambda: ops.colmean_f16(lin(x), gamma, beta, 1e-5, resid32=y32)
    a, b = two().clone(), fused().clone()
    d = (a - b).abs()
    print(f"N={n} S={S}: max |diff| {float(d.max()):.3e} (values up to {float(a.abs().max()):.3f}), "
          f"linear + colmean {timed(two):.1f} us, fused + finish {timed(fused):.1f} us")

    out2 = torch.empty_like(out)

    def fused_ffn():
        st = ffn(x.data_ptr(), lin1.w.data_ptr(), lin1.b.data_ptr(), lin.w.data_ptr(), lin.b.data_ptr(), y32.data_ptr(), gamma.data_ptr(),
                 beta.data_ptr(), 1e-5, out2.data_ptr(), ws.data_ptr(), ws.numel() * 4, n, S, torch.cuda.current_stream().cuda_stream)
        assert st == 0, L.fp_last_error()
        return out2

    three = lambda: ops.colmean_f16(lin(lin1(x, relu=True)), gamma, beta, 1e-5, resid32=y32)
    a, b = three().clone(), fused_ffn().clone()
    d = (a - b).abs()
    print(f"          FFN: max |diff| {float(d.max()):.3e}, linear1 + linear2 + colmean {timed(three):.1f} us, fused + finish {timed(fused_ffn):.1f} us")
