"""phase timers of k_rows512 (csrc/linear_ln.hip, profiling build): first loop incl. cold start / park + second loop (FFN form) / park of the output /
LayerNorm rows, per workgroup, 100 MHz wall clock, wave 0
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so python scripts/dbg_linear_ln.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops

dev = torch.device("cuda:0")
L = C.CDLL(os.environ["FP_AMD_LIB"])
g = torch.Generator(device="cpu").manual_seed(4)
r = lambda *s, k=1.0: (torch.randn(s, generator=g) * k).to(dev)
w0, b0, w1, b1, w2, b2 = r(512, 512, k=0.05).half(), r(512, k=0.1), r(512, 512, k=0.05).half(), r(512, k=0.1), r(512, 512, k=0.05).half(), r(512, k=0.1)
p0, p1, p2 = ops.PackedLinear512(w0), ops.PackedLinear512(w1), ops.PackedLinear512(w2)
gamma, beta = 1.0 + r(512, k=0.1), r(512, k=0.1)
pe = r(400, 512)


def timed(f, reps=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


only = os.environ.get("FP_FORMS", "")           # e.g. FP_FORMS=FFN: that form at N = 126 only
for n in ((126,) if only else (126, 252, 160)):
    x = r(n, 400, 512).half()
    tok = r(n, 400, 512).half()
    y32 = r(n, 400, 512)
    forms = {"out_proj + LN (tokens + pe -> y32, y16)": lambda: ops.linear_layernorm_res(x, p0, b0, gamma, beta, tok16=tok, pe=pe),
             "out_proj + LN (x32 -> y32, y16)": lambda: ops.linear_layernorm_res(x, p0, b0, gamma, beta, x32=y32),
             "FFN + LN + mean": lambda: ops.ffn_layernorm_mean(x, p1, b1, p2, b2, y32, gamma, beta)}
    for name, f in forms.items():
        if only and only not in name:
            continue
        us = timed(f)
        out = (C.c_ulonglong * 8)()
        L.fp_dbg_linear_ln(out, 1)
        f(); torch.cuda.synchronize()
        L.fp_dbg_linear_ln(out, 0)
        t = out[4]
        ph = [out[i] / t * 10 / 1e3 for i in range(4)]
        print(f"N={n} {name}: {us:.1f} us; {t} tiles = {t / 256:.2f} rounds; per tile: loop1 {ph[0]:.1f} us, park+loop2 {ph[1]:.1f} us, park {ph[2]:.1f} us, "
              f"LN rows {ph[3]:.1f} us, sum {sum(ph):.1f} us; kernel span {(out[7] - out[6]) * 10 / 1e3:.1f} us", flush=True)
