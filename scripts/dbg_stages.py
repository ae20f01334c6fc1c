import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
import bench
from foundationpose_amd import ops
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor, make_crop_data_batch
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
dev = torch.device('cuda:0')
N = 252
sc = bench.build_scene(dev, 0, N)
ref = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev, precision="fp16")
plan = ref.plan()
rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
depth_t = torch.as_tensor(sc["depth"], device=dev)
xyz_t = ops.depth_to_xyz(depth_t, sc["K"], f64_internal=True)
poses0 = torch.as_tensor(sc["poses"], device=dev)
AB = torch.empty((2 * N, 6, 160, 160), dtype=torch.float16, device=dev)
def T(name, fn, n=2):
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
    return r
batch = T("crop_data_batch", lambda: make_crop_data_batch([160, 160], poses0, sc["mesh"], rgb_t, depth_t, sc["K"], 1.2, xyz_t, cfg=ref.cfg, mesh_tensors=sc["gm"], mesh_diameter=sc["diameter"], AB=AB))
enc = plan.enc
x = T("conv1_hip", lambda: ops.conv7x7s2_bn_relu(AB, enc.c1_wflat, enc.c1.scale, enc.c1.shift, channels_last=enc.cl))
x2 = T("c2", lambda: enc.c2(x))
x3 = T("s2", lambda: enc.s2(x2))
x4 = T("s3", lambda: enc.s3(x3))
ab = T("cat", lambda: torch.cat((x4[:N], x4[N:]), dim=1))
print(ab.shape, ab.stride(), ab.is_contiguous(memory_format=torch.channels_last))
y = T("j0", lambda: enc.j0(ab))
y = T("j1", lambda: enc.j1(y))
y = T("j2", lambda: enc.j2(y))
y = T("j3", lambda: enc.j3(y))
y = T("j4", lambda: enc.j4(y))
tok = T("tokens", lambda: (y.permute(0, 2, 3, 1).reshape(N, -1, y.shape[1]) + enc.pe[:, :400]))
layer, w, b = plan.heads["trans"]
q = T("qkv", lambda: layer.att.qkv(tok))
a = T("att", lambda: layer.att(tok))
h = T("layer", lambda: layer(tok))
o = T("head", lambda: F.linear(h, w, b).float().mean(dim=1))
T("full_plan", lambda: plan(AB))
T("predict5", lambda: ref.predict(rgb_t, depth_t, sc["K"], poses0, xyz_t, mesh=sc["mesh"], mesh_tensors=sc["gm"], mesh_diameter=sc["diameter"], iteration=5), n=2)
