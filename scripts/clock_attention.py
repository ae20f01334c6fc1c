"""shader clock and socket power (amdgpu hwmon, bench.ClockSampler) while fp_attention_f16_fwd runs back to back for ~0.5 s:
    python scripts/clock_attention.py            # FP_AMD_LIB=... for a variant"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ClockSampler
from foundationpose_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
qkv = (torch.randn((126, 400, 1536), device=dev) * 1.5).half()
for _ in range(20):
    ops.attention_f16(qkv, 4)
torch.cuda.synchronize()
cs = ClockSampler(0, period_s=0.002)
with cs:
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.6:
        for _ in range(100):
            ops.attention_f16(qkv, 4)
        torch.cuda.synchronize()
        n += 100
    dt = time.perf_counter() - t0
rep = cs.summary()
print(json.dumps(dict(lib=os.path.basename(os.environ.get("FP_AMD_LIB", "libfp_amd.so")), us_per_launch=round(dt / n * 1e6, 1), clock=rep)))
