"""BASELINE configs[4]: tracking mode -- a 1000-frame synthetic RGB-D sequence of a moving object,
64 pose hypotheses per frame (perturbations of the previous frame's pose), 2 refine iterations, captured hipGraphs
replayed per frame (foundationpose_amd/graphs.py: depth erosion + bilateral filter + back-projection + 2 x (crop windows,
rasteriser, observed crop, RefineNet, pose update)).  Every frame is different: its own GT pose on a smooth trajectory,
rendered by the product's rasteriser, with its own noise / dropout; per frame the uint8 colour image, the float depth
map and the hypotheses are uploaded from pinned host memory (the H2D copies are inside the timed region).
Round 5: the measurement itself lives in bench.py (`tracking_bench`, also part of the driver's `python bench.py` record,
key `tracking`); this is the stand-alone command line over it, also for the reference's own track_one (--hyps 1).

Also measures the register() refine loop (252 hypotheses x 5 iterations) as ONE hipGraph against eager launches.
Prints one JSON line (secondary metric; bench.py keeps the headline)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
ap.add_argument("--no-register-graph", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
sc = bench.build_scene(dev, 0, 252)
refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
seq = bench.make_sequence(dev, sc, args.frames)
out = {"metric": "tracking frames/sec (BASELINE configs[4]: 1000-frame synthetic RGB-D sequence, 64 hyp/frame, 2 refine iterations, hipGraph replay)",
       "sequence_generation_s": seq[3]}
out.update(bench.tracking_bench(dev, sc, refiner, seq, args.hyps, args.iters))

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
