"""phase timers of k_attention_f16 (profiling build): prologue / block loop / output per workgroup, 100 MHz wall clock, thread 0
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so python scripts/dbg_attention.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops

dev = torch.device("cuda:0")
L = C.CDLL(os.environ["FP_AMD_LIB"])
torch.manual_seed(0)
for B in (126, 252):
    qkv = (torch.randn((B, 400, 1536), device=dev) * 1.5).half()
    for f16s in (False, True):
        fn = lambda: ops.attention_f16(qkv, 4, fp16_scores=f16s)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out = (C.c_ulonglong * 8)()
        L.fp_dbg_attention(out, 1)
        fn(); torch.cuda.synchronize()
        L.fp_dbg_attention(out, 0)
        n = out[3]
        ph = [out[i] / n * 10 / 1e3 for i in range(3)]
        print(f"B={B} fp16_scores={f16s}: {us:.1f} us; {n} workgroups = {n / 256:.2f} rounds; per workgroup: prologue {ph[0]:.1f} us, loop {ph[1]:.1f} us "
              f"({ph[1] / 7:.2f} per block), output {ph[2]:.1f} us, sum {sum(ph):.1f} us; kernel span {(out[7] - out[6]) * 10 / 1e3:.1f} us", flush=True)
