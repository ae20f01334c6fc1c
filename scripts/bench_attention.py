"""fp_attention_f16_fwd at the bench shapes (S=400, 4 heads of 128): HIP events, both score policies.  A/B builds: FP_AMD_LIB=..."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops

dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("FP_AMD_LIB", "libfp_amd.so"))
for B in (126, 252):
    qkv = (torch.randn((B, 400, 1536), device=dev) * 1.5).half()
    for f16s in (False, True):
        fn = lambda: ops.attention_f16(qkv, 4, fp16_scores=f16s)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(json.dumps(dict(lib=tag, B=B, fp16_scores=f16s, us=round(ms * 1e3, 1), TFLOPs=round(B * 4 * 4 * 400 * 400 * 128 / ms / 1e9, 1))), flush=True)
