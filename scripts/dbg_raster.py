"""debug: the rasteriser under a concurrent MFMA kernel on another stream -- what differs?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from foundationpose_amd import synthetic as syn, ops
from foundationpose_amd.Utils import get_mesh_handle
dev = torch.device("cuda:0")
sc = bench.build_scene(dev, 0, 252)
n = 38
P = torch.as_tensor(sc["poses"][:n], device=dev)
h = get_mesh_handle(sc["gm"])
K, diam = sc["K"], sc["diameter"]
ws = torch.empty(max(16, ops.workspace_bytes(n, h.V, h.T, 160, 160)), dtype=torch.uint8, device=dev)
tf, bb = ops.crop_windows(P, K, diam, 1.2, (160, 160))
side = torch.cuda.Stream(device=dev)
g = torch.Generator(device="cpu").manual_seed(0)
M = 14800
x = (torch.randn((M, 512), generator=g) * 0.1).half().to(dev)
w = (torch.randn((512, 512), generator=g) * 0.05).half().to(dev)
b0 = torch.zeros(512, device=dev)
y = torch.empty((M, 512), dtype=torch.float16, device=dev)
Gm = ops.IgemmGeom.matrix
hog = sys.argv[1] if len(sys.argv) > 1 else "lin"
big = torch.empty(64 << 20, dtype=torch.float32, device=dev)
F16 = len(sys.argv) > 2 and sys.argv[2] == "f16"
NHOG = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ORDER = sys.argv[4] if len(sys.argv) > 4 else "rasterfirst"
qkv = (torch.randn((37, 400, 1536), generator=g) * 0.3).half().to(dev)
AB = torch.zeros((n, 6, 160, 160), dtype=torch.float16 if F16 else torch.float32, device=dev)
def render(out_f16):
    r = ops.render_crops(h, P, bb, K, syn.H, syn.W, out_hw=(160, 160), mesh_diameter=diam, xyz_thr=0.001, normalize_xyz=True,
                         A_out=AB, workspace=ws, want=("A", "zbuf", "tri_id"))
    return r
with torch.inference_mode():
    base = {k: v.clone() for k, v in render(False).items()}
    torch.cuda.synchronize()
    nbad = 0
    for rep in range(60):
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        def hogs():
            with torch.cuda.stream(side):
                for _ in range(NHOG):
                    if hog == "lin":
                        ops.igemm_f16(x, Gm(512), w, b0, y, Gm(512), M, 512, 512, 1, relu=False)
                    elif hog == "copy":
                        big.mul_(1.0001)
                    elif hog == "matmul":
                        torch.matmul(x, w.t(), out=y)
                    elif hog == "attn":
                        ops.attention_f16(qkv, 4)
        if ORDER == "hogfirst":
            hogs()
        r = render(False)
        if ORDER != "hogfirst":
            hogs()
        main.wait_stream(side)
        torch.cuda.synchronize()
        diffs = {k: int((r[k] != base[k]).sum().item()) for k in r}
        if any(diffs.values()):
            nbad += 1
            if nbad <= 3:
                print("rep", rep, diffs, flush=True)
                d = (r["A"] != base["A"]).any(dim=1)          # (n, 160, 160)
                idx = torch.nonzero(d)
                print("  pixels:", idx[:12].tolist(), flush=True)
                for (hh, j, i) in idx[:4].tolist():
                    print(f"   hyp {hh} px ({j},{i}) tri base {int(base['tri_id'][hh, j, i])} now {int(r['tri_id'][hh, j, i])}  A base {[round(float(v), 4) for v in base['A'][hh, :, j, i]]} now {[round(float(v), 4) for v in r['A'][hh, :, j, i]]}", flush=True)
                    t = int(base['tri_id'][hh, j, i])
                    # all pixels of that triangle
                    same_tri = (base['tri_id'][hh] == t)
                    print(f"     pixels of tri {t}: {int(same_tri.sum())}, differing among them {int((d[hh] & same_tri).sum())}; differing pixels in hyp {int(d[hh].sum())}, triangles involved {torch.unique(base['tri_id'][hh][d[hh]]).tolist()[:10]}", flush=True)
    print(f"hog={hog} f16={F16} nhog={NHOG} order={ORDER}: mismatching runs {nbad} of 60", flush=True)
