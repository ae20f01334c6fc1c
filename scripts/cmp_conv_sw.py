"""sha1 of the outputs of the stride-1 3x3 convs at several shapes (full / partial last tile, residual, BatchNorm, token
layout + positional output): run once per build / mode and compare the lines (profiling aid:
FP_AMD_LIB=.../libfp_amd_profile.so FP_CONV_SW=classic python scripts/cmp_conv_sw.py)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_amd import ops
dev = torch.device("cuda:0")
G = ops.IgemmGeom.image
g = torch.Generator(device="cpu").manual_seed(1)
def sha(t):
    return hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
for (B, H, Ci, Co, res, bn) in ((252, 40, 128, 128, True, True), (126, 40, 256, 256, True, False), (126, 20, 512, 512, True, True), (37, 40, 256, 256, False, True),
                               (75, 20, 512, 512, True, False), (3, 40, 128, 128, True, True), (5, 20, 512, 512, False, False), (504, 40, 128, 128, False, False)):
    x = torch.zeros((B, H + 2, H + 2, Ci), dtype=torch.float16)
    x[:, 1:-1, 1:-1] = torch.relu(torch.randn((B, H, H, Ci), generator=g) * 0.5).half()
    x = x.to(dev)
    w = (torch.randn((Co, 9 * Ci), generator=g) * 0.02).half().to(dev)
    b = torch.randn(Co, generator=g).half().float().to(dev)
    sc, sh = (torch.rand(Co, generator=g) + 0.5).to(dev), (torch.randn(Co, generator=g) * 0.1).to(dev)
    r = (torch.randn((B, H + 2, H + 2, Co), generator=g) * 0.5).half().to(dev)
    y = torch.zeros((B, H + 2, H + 2, Co), dtype=torch.float16, device=dev)
    M = B * H * H
    wt = ops.pack_conv3x3_tiles(w, Co, Ci) if os.environ.get("FP_W_TILES") == "1" else None      # FP_W_TILES=1: the tile-packed weight path
    for rep in range(2):
        y.zero_()
        ops.igemm_f16(x, G(H, H, 1, Ci, stride=1, offset=0), w, b, y, G(H, H, 1, Co), M, Co, Ci, 9, relu=True, residual=r if res else None,
                      r_geom=G(H, H, 1, Co) if res else None, bn_scale=sc if bn else None, bn_shift=sh if bn else None, conv_rounding=True, w_tiles=wt)
    print(f"conv B={B} H={H} {Ci}->{Co} res={res} bn={bn}: {sha(y)} border0={float(y[:, 0].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0}", flush=True)
    if Co == 512:
        tok = torch.zeros((B, H * H, Co), dtype=torch.float16, device=dev); tp = torch.zeros_like(tok)
        pe = torch.randn((H * H, Co), generator=g).to(dev)
        ops.igemm_f16(x, G(H, H, 1, Ci, stride=1, offset=0), w, b, tok, G(H, H, 0, Co), M, Co, Ci, 9, relu=True, residual=r if res else None,
                      r_geom=G(H, H, 1, Co) if res else None, conv_rounding=True, pe=pe, y_pe=tp, w_tiles=wt)
        print(f"  tokens: {sha(tok)} {sha(tp)}", flush=True)
