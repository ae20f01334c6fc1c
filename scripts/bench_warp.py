"""fp_warp_crops at the bench shapes + edge cases: time (HIP events) and a digest of every output, so that two builds / kernel
variants can be compared bit for bit:
    python scripts/bench_warp.py                                                             # product (k_warp2: wave-private LDS row segments)
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so FP_WARP_V=1 python scripts/bench_warp.py   # the gather kernel k_warp"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from foundationpose_amd import ops

dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("FP_AMD_LIB", "libfp_amd.so")) + (" FP_WARP_V=" + os.environ["FP_WARP_V"] if "FP_WARP_V" in os.environ else "")
N = 126
sc = bench.build_scene(dev, 0, N)
rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
depth_t = ops.bilateral_filter_depth(ops.erode_depth(torch.as_tensor(sc["depth"], device=dev)))
xyz_t = ops.depth_to_xyz(depth_t, sc["K"], f64_internal=True)
P = torch.as_tensor(sc["poses"], device=dev).clone()
P[:, :3, 3] += torch.linspace(-0.01, 0.01, N, device=dev)[:, None]
# edge cases: far (window smaller than the crop: up-sampling), near (window ~3.5 x the crop), very near (segment > cap: gather path),
# windows leaving the frame on each side
E = P[:8].clone()
E[0, 2, 3] *= 2.5; E[1, 2, 3] *= 0.45; E[2, 2, 3] *= 0.2; E[3, 0, 3] -= 0.22; E[4, 0, 3] += 0.25; E[5, 1, 3] -= 0.2; E[6, 1, 3] += 0.2; E[7, 2, 3] *= 0.3
out = {}
for name, poses in (("bench", P), ("edge", E)):
    tf, _ = ops.crop_windows(poses, sc["K"], sc["diameter"], 1.2, (160, 160))
    n = poses.shape[0]
    for mode, nm in ((ops.MODE_REFINE, "refine"), (ops.MODE_SCORE, "score")):
        for f16 in (True, False):
            B = torch.zeros((n, 6, 160, 160), dtype=torch.float16 if f16 else torch.float32, device=dev)
            fn = lambda: ops.warp_crops(rgb_t, xyz_t if mode == ops.MODE_REFINE else None, depth_t if mode == ops.MODE_SCORE else None, tf, sc["K"],
                                        poses, sc["diameter"], mode, normalize_xyz=True, out_hw=(160, 160), B_out=B)
            fn(); torch.cuda.synchronize()
            key = f"{name}_{nm}_{'f16' if f16 else 'f32'}"
            out[key] = hashlib.sha1(B.cpu().numpy().tobytes()).hexdigest()[:12]
            if name == "bench" and f16:
                for _ in range(5): fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(40): fn()
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 40 * 1e3)
                out[f"us_{nm}"] = round(sorted(ts)[2], 2)
# a small odd-sized crop (one partial wave) and a crop wider than three waves (falls back to k_warp)
for ow_, oh_ in ((40, 24), (200, 160)):
    tf, _ = ops.crop_windows(P[:4], sc["K"], sc["diameter"], 1.2, (ow_, oh_))
    B = ops.warp_crops(rgb_t, xyz_t, None, tf, sc["K"], P[:4], sc["diameter"], ops.MODE_REFINE, normalize_xyz=False, out_hw=(oh_, ow_))
    out[f"crop_{ow_}x{oh_}"] = hashlib.sha1(B.cpu().numpy().tobytes()).hexdigest()[:12]
print("WARP " + json.dumps(dict(lib=tag, **out)), flush=True)
