"""BASELINE configs[4]: tracking mode -- a 1000-frame synthetic RGB-D sequence of a moving object,
64 pose hypotheses per frame (perturbations of the previous frame's pose), 2 refine iterations, captured hipGraphs
replayed per frame (foundationpose_amd/graphs.py: depth erosion + bilateral filter + back-projection + 2 x (crop windows,
rasteriser, observed crop, RefineNet, pose update)).  Every frame is different: its own GT pose on a smooth trajectory,
rendered by the product's rasteriser, with its own noise / dropout; per frame the uint8 colour image, the float depth
map and the 64 hypotheses are uploaded from pinned host memory (the H2D copies are inside the timed region).

Also measures the register() refine loop (252 hypotheses x 5 iterations) as ONE hipGraph against eager launches.
Prints one JSON line (secondary metric; bench.py keeps the headline)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from foundationpose_amd import synthetic as syn
from foundationpose_amd.Utils import nvdiffrast_render
from foundationpose_amd.graphs import GraphedTracker
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--hyps", type=int, default=64)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--no-register-graph", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
F_, N, R = args.frames, args.hyps, args.iters
sc = bench.build_scene(dev, 0, 252)
refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)

# ---- the sequence: GT pose on a smooth trajectory (<= 4 mm and <= 1.5 deg per frame), rendered frame by frame on the GPU
rng = np.random.default_rng(7)
T0 = sc["T"].astype(np.float64)
gt = np.zeros((F_, 4, 4))
for f in range(F_):
    a = 2 * np.pi * f / 250.0
    ax = np.array([np.sin(0.7 * a), np.cos(a), 0.3])
    ax /= np.linalg.norm(ax)
    ang = np.deg2rad(20.0) * np.sin(a)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    gt[f] = T0
    gt[f, :3, :3] = dR @ T0[:3, :3]
    gt[f, :3, 3] = T0[:3, 3] + np.array([0.06 * np.sin(a), 0.04 * np.sin(2 * a), 0.05 * np.cos(a) - 0.05])
rgb_h = torch.empty((F_, syn.H, syn.W, 3), dtype=torch.uint8).pin_memory()
depth_h = torch.empty((F_, syn.H, syn.W), dtype=torch.float32).pin_memory()
gen = torch.Generator(device=dev).manual_seed(11)
bg = torch.as_tensor(np.kron(rng.uniform(0.3, 0.6, size=(syn.H // 8, syn.W // 8, 3)), np.ones((8, 8, 1))), device=dev, dtype=torch.float32)
t_gen = time.perf_counter()
for f in range(F_):
    color, depth, _ = nvdiffrast_render(K=sc["K"], H=syn.H, W=syn.W, ob_in_cams=torch.as_tensor(gt[f][None], device=dev, dtype=torch.float),
                                        mesh_tensors=sc["gm"], use_light=True, extra={})
    mask = depth[0] > 0
    rgb = torch.where(mask[..., None], color[0], bg)
    d = torch.where(mask, depth[0], torch.full_like(depth[0], 1.2)) + torch.randn((syn.H, syn.W), generator=gen, device=dev) * 0.001
    d = torch.where(torch.rand((syn.H, syn.W), generator=gen, device=dev) < 0.02, torch.zeros_like(d), d)
    rgb_h[f].copy_((rgb.clamp(0, 1) * 255).to(torch.uint8))
    depth_h[f].copy_(d)
torch.cuda.synchronize()
t_gen = time.perf_counter() - t_gen
# hypotheses of frame f: 64 perturbations (<= 2 cm, <= 10 deg) of the pose of frame f-1
hyp_h = torch.empty((F_, N, 4, 4), dtype=torch.float32).pin_memory()
for f in range(F_):
    hyp_h[f].copy_(torch.from_numpy(syn.perturbed_poses(gt[max(f - 1, 0)], N, seed=100 + f, max_trans=0.02, max_rot_deg=10.0).astype(np.float32)))

trk = GraphedTracker(refiner, sc["gm"], sc["diameter"], sc["K"], syn.H, syn.W, n_hyp=N, iteration=R, device=dev).capture()
rgb_u8 = torch.empty((syn.H, syn.W, 3), dtype=torch.uint8, device=dev)


def frame(f, graph=True):
    rgb_u8.copy_(rgb_h[f], non_blocking=True)            # H2D, 0.92 MB
    trk.rgb.copy_(rgb_u8)                                # u8 -> f32 on the device
    trk.depth.copy_(depth_h[f], non_blocking=True)       # H2D, 1.2 MB
    trk.poses_in.copy_(hyp_h[f], non_blocking=True)      # H2D, 4 KB
    return trk.replay() if graph else trk._body()


res = {}
with torch.inference_mode():
    for name, graph in (("hipgraph", True), ("eager", False)):
        for f in range(5):
            frame(f, graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(F_):
            out = frame(f, graph)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / F_
    # per-frame latency (host waits for every frame, as a control loop would)
    lat = []
    for f in range(min(F_, 200)):
        t0 = time.perf_counter()
        frame(f, True)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    # sanity: the refined hypotheses of the last frame moved towards that frame's pose (untrained stand-in weights: finite is all we ask)
    assert torch.isfinite(out).all()

out = {"metric": "tracking frames/sec (BASELINE configs[4]: 1000-frame synthetic RGB-D sequence, 64 hyp/frame, 2 refine iterations, hipGraph replay)",
       "frames": F_, "hypotheses_per_frame": N, "refine_iterations": R, "distinct_frames": F_,
       "uploads_per_frame_bytes": int(rgb_h[0].numel() + depth_h[0].numel() * 4 + hyp_h[0].numel() * 4),
       "sequence_generation_s": t_gen,
       "hipgraph_ms_per_frame": res["hipgraph"] * 1e3, "eager_ms_per_frame": res["eager"] * 1e3,
       "frames_per_sec": 1.0 / res["hipgraph"], "hypothesis_passes_per_sec": N * R / res["hipgraph"],
       "speedup_vs_eager": res["eager"] / res["hipgraph"],
       "latency_ms_synced_per_frame": {"median": float(np.median(lat) * 1e3), "p95": float(np.percentile(lat, 95) * 1e3)}}

if not args.no_register_graph:
    # the register() refine loop (estimater.py:215: 252 hypotheses, 5 iterations, incl. the depth filters) as ONE graph
    reg = GraphedTracker(refiner, sc["gm"], sc["diameter"], sc["K"], syn.H, syn.W, n_hyp=252, iteration=5, device=dev).capture()
    P = torch.as_tensor(sc["poses"], device=dev)
    rgb_f, dep = torch.as_tensor(sc["rgb"], device=dev).float(), torch.as_tensor(sc["depth"], device=dev)
    tt = {}
    with torch.inference_mode():
        for name, fn in (("eager", lambda: reg.step_eager(rgb_f, dep, P)), ("hipgraph", lambda: reg.step(rgb_f, dep, P))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            tt[name] = (time.perf_counter() - t0) / 10
    out["register_refine_loop_252x5"] = {"eager_ms": tt["eager"] * 1e3, "hipgraph_ms": tt["hipgraph"] * 1e3,
                                         "delta_ms": (tt["eager"] - tt["hipgraph"]) * 1e3}
print(json.dumps(out))
