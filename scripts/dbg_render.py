import sys; sys.path.insert(0, '.')
import numpy as np, torch
from foundationpose_amd import ops, synthetic as syn
from foundationpose_amd.mesh import make_can_mesh
from foundationpose_amd.Utils import make_mesh_tensors, sample_views_icosphere, euler_matrix
from oracle import ops as oo, pipeline as op
dev = torch.device('cuda:0')
mesh = make_can_mesh(); mnp = op.mesh_tensors_np(mesh); gm = make_mesh_tensors(mesh, device=dev)
K = syn.YCBV_K; T = syn.gt_pose(0)
cams = sample_views_icosphere(40)
grid = np.asarray([np.linalg.inv(c @ euler_matrix(0,0,a)) for c in cams for a in np.deg2rad(np.arange(0,360,60))])
grid[:, :3, 3] = T[:3, 3]; P = grid[::7].astype(np.float32)
diam = 0.1737
tf, bb = oo.crop_windows(P, K, diam, 1.2)
want = ("A","color","depth","xyz","normal")
ref = oo.render_crops(mnp, P, bb, K, 480, 640, (160,160), diam, 0.1, True, want=want)
out = ops.render_crops(gm["_handle"], torch.as_tensor(P, device=dev), torch.as_tensor(bb, device=dev), K, 480, 640, (160,160), diam, 0.1, True, want=want)
for k in want:
    o = out[k].cpu().numpy(); r = ref[k]
    neq = (o != r)
    print(k, "mismatch frac", neq.mean(), "max abs", np.abs(o - r).max())
    if neq.any():
        idx = np.argwhere(neq)[0]; print("  first", idx, o[tuple(idx)], r[tuple(idx)])
A16 = ops.render_crops(gm["_handle"], torch.as_tensor(P, device=dev), torch.as_tensor(bb, device=dev), K, 480, 640, (160,160), diam, 0.1, True, out_f16=True, want=("A",))["A"].cpu().numpy()
r16 = ref["A"].astype(np.float16)
print("fp16 mismatch", (A16 != r16).mean(), np.abs(A16.astype(np.float32) - r16.astype(np.float32)).max())
