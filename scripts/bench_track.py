"""BASELINE configs[4]: tracking mode, synthetic RGB-D sequence, 64 hypotheses per frame, 2 refine iterations,
hipGraph replay vs eager launches.  Prints one JSON line (secondary metric; bench.py keeps the headline)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from foundationpose_amd import synthetic as syn
from foundationpose_amd.graphs import GraphedTracker
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--hyps", type=int, default=64)
ap.add_argument("--iters", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
sc = bench.build_scene(dev, 0, args.hyps)
refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
P = syn.perturbed_poses(sc["T"], args.hyps, seed=3, max_trans=0.02, max_rot_deg=10.0).astype(np.float32)
trk = GraphedTracker(refiner, sc["gm"], sc["diameter"], sc["K"], syn.H, syn.W, n_hyp=args.hyps, iteration=args.iters, device=dev).capture()
rgb = torch.as_tensor(sc["rgb"], device=dev).float()
depth = torch.as_tensor(sc["depth"], device=dev)
Pd = torch.as_tensor(P, device=dev)
res = {}
for name, fn in (("eager", lambda: trk.step_eager(rgb, depth, Pd)), ("hipgraph", lambda: trk.step(rgb, depth, Pd))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        fn()
    torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t0) / args.frames
print(json.dumps({"metric": "tracking frames/sec (64 hyp/frame, 2 refine iterations, 640x480 RGB-D, incl. depth filtering)",
                  "frames": args.frames, "hypotheses_per_frame": args.hyps, "refine_iterations": args.iters,
                  "eager_ms_per_frame": res["eager"] * 1e3, "hipgraph_ms_per_frame": res["hipgraph"] * 1e3,
                  "frames_per_sec": 1.0 / res["hipgraph"], "hypothesis_passes_per_sec": args.hyps * args.iters / res["hipgraph"],
                  "speedup_vs_eager": res["eager"] / res["hipgraph"]}))
