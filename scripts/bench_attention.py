"""fp_attention_f16_fwd at the bench shapes (S=400, 4 heads of 128): HIP events, both score policies.  A/B builds: FP_AMD_LIB=..."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops

dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("FP_AMD_LIB", "libfp_amd.so")) + "".join(f" {k}={os.environ[k]}" for k in ("FP_ATT_QT", "FP_ATT_WAVES", "FP_ATT_PP") if k in os.environ)
torch.manual_seed(0)
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
        dg = hashlib.sha1(fn().cpu().numpy().tobytes()).hexdigest()[:12]      # variants of the kernel must agree bit for bit
        print(json.dumps(dict(lib=tag, B=B, fp16_scores=f16s, us=round(ms * 1e3, 1), TFLOPs=round(B * 4 * 4 * 400 * 400 * 128 / ms / 1e9, 1), digest=dg)), flush=True)
for B, S in ((3, 130), (2, 33), (1, 400), (5, 64)):      # ragged sizes: digests only
    qkv = (torch.randn((B, S, 1536), device=dev) * 1.5).half()
    print(json.dumps(dict(lib=tag, B=B, S=S, digest=[hashlib.sha1(ops.attention_f16(qkv, 4, fp16_scores=f).cpu().numpy().tobytes()).hexdigest()[:12] for f in (False, True)])), flush=True)
