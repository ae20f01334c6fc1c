"""Launches every hand-written kernel of libfp_amd.so at the bench sizes (N=252) a few times: the target command of the
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in separate passes, see scripts/pmc_traffic.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from foundationpose_amd import ops

dev = torch.device("cuda:0")
N = int(os.environ.get("FP_N", "252"))
reps = int(os.environ.get("FP_REPS", "3"))
sc = bench.build_scene(dev, 0, N)
h = sc["gm"]["_handle"]
rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
depth_t = ops.bilateral_filter_depth(ops.erode_depth(torch.as_tensor(sc["depth"], device=dev)))
xyz_t = ops.depth_to_xyz(depth_t, sc["K"], f64_internal=True)
poses = torch.as_tensor(sc["poses"], device=dev)
AB = torch.empty((2 * N, 6, 160, 160), dtype=torch.float16, device=dev)
g = torch.Generator(device="cpu").manual_seed(0)
w1 = (torch.randn((64, 294), generator=g) * 0.05).half().to(dev)
b1, sc1, sh1 = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev)
P1 = torch.zeros((2 * N, 82, 82, 64), dtype=torch.float16, device=dev)
x = torch.randn((N * 400, 512), generator=g).half().to(dev)
wq = (torch.randn((1536, 512), generator=g) * 0.05).half().to(dev)
bq = torch.zeros(1536, device=dev)
q = torch.empty((N * 400, 1536), dtype=torch.float16, device=dev)
G = ops.IgemmGeom
xi = torch.relu(torch.randn((N, 42, 42, 256), generator=g) * 0.5).half().to(dev)
wi = (torch.randn((256, 9 * 256), generator=g) * 0.02).half().to(dev)
bi = torch.zeros(256, device=dev)
yi = torch.zeros((N, 42, 42, 256), dtype=torch.float16, device=dev)
lnw, lnb = torch.ones(512, device=dev), torch.zeros(512, device=dev)
pe = torch.randn((400, 512), generator=g).to(dev)
xt = x.reshape(N, 400, 512)
for _ in range(reps):
    tf, bb = ops.crop_windows(poses, sc["K"], sc["diameter"], 1.2, (160, 160))
    ops.render_crops(h, poses, bb, sc["K"], 480, 640, (160, 160), sc["diameter"], 0.001, True, A_out=AB[:N])
    ops.warp_crops(rgb_t, xyz_t, None, tf, sc["K"], poses, sc["diameter"], ops.MODE_REFINE, True, B_out=AB[N:])
    ops.warp_crops(rgb_t, None, depth_t, tf, sc["K"], poses, sc["diameter"], ops.MODE_SCORE, True, B_out=AB[N:])
    ops.conv7x7s2_bn_relu(AB, w1, b1, sc1, sh1, P1, 1)
    ops.igemm_f16(x, G.matrix(512), wq, bq, q, G.matrix(1536), N * 400, 1536, 512, 1)                      # in_proj (k_igemm_pp)
    ops.igemm_f16(xi, G.image(40, 40, 1, 256, offset=0), wi, bi, yi, G.image(40, 40, 1, 256), N * 1600, 256, 256, 9, relu=True,
                  conv_rounding=True)                                                                          # 256->256 conv (k_conv_sw)
    ops.attention_f16(q.reshape(N, 400, 1536), 4)
    ops.add_pe_f16(xt, pe)
    y32, y16 = ops.layernorm_res(xt, lnw, lnb, 1e-5, tok16=xt, pe=pe)
    ops.colmean_f16(xt, lnw, lnb, 1e-5, resid32=y32)
torch.cuda.synchronize()
print("ok")
