"""bitwise comparison of two builds / variants of fp_attention_f16_fwd: run once per variant with OUT=<file>, then with A=<file> B=<file>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if "OUT" in os.environ:
    import torch
    from foundationpose_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    res = {}
    for B, S in ((2, 400), (2, 130), (1, 128), (1, 192)):
        qkv = (torch.randn((B, S, 1536), device=dev) * 1.5).half()
        for f in (False, True):
            res[f"{B}x{S}_{int(f)}"] = ops.attention_f16(qkv, 4, fp16_scores=f).float().cpu().numpy()
    np.savez(os.environ["OUT"], **res)
else:
    a, b = np.load(os.environ["A"]), np.load(os.environ["B"])
    for k in a.files:
        d = np.abs(a[k] - b[k])
        nz = np.argwhere(d > 0)
        print(k, "differing", len(nz), "of", d.size, "max", d.max(), "first", nz[:5].tolist(), "queries", sorted(set(nz[:, 1].tolist()))[:20] if len(nz) else [])
