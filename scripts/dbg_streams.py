"""debug: sub-batches on streams -- which rows differ, by how much, in which mode"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from foundationpose_amd import synthetic as syn
from foundationpose_amd.graphs import GraphedTracker
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict
dev = torch.device("cuda:0")
sc = bench.build_scene(dev, 0, 252)
NH = int(sys.argv[1]) if len(sys.argv) > 1 else 75
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P = torch.as_tensor(sc["poses"][:NH], device=dev)
rgb, dep = torch.as_tensor(sc["rgb"], device=dev).float(), torch.as_tensor(sc["depth"], device=dev)
def mk(ns):
    return PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev, n_streams=ns)
r1, r2 = mk(1), mk(2)
t1 = GraphedTracker(r1, sc["gm"], sc["diameter"], sc["K"], syn.H, syn.W, n_hyp=NH, iteration=IT, device=dev).capture()
t2 = GraphedTracker(r2, sc["gm"], sc["diameter"], sc["K"], syn.H, syn.W, n_hyp=NH, iteration=IT, device=dev).capture()
ref = t1.step_eager(rgb, dep, P).clone()
def rep(name, x):
    d = (x - ref).abs().amax(dim=(1, 2)).cpu().numpy()
    bad = np.nonzero(d > 0)[0]
    print(f"{name:28s} rows differing {len(bad):3d}  max {d.max():.3e}  first rows {bad[:8].tolist()} parts {r2.sub.parts(NH)}", flush=True)
bad = dict(g1=0, e2=0, g2=0)
REPS = 40
for k in range(REPS):
    bad["g1"] += int(not torch.equal(t1.step(rgb, dep, P), ref))
    bad["e2"] += int(not torch.equal(t2.step_eager(rgb, dep, P), ref))
    bad["g2"] += int(not torch.equal(t2.step(rgb, dep, P), ref))
print(f"N={NH} it={IT} parts={r2.sub.parts(NH)}: mismatching runs of {REPS}: graph 1 part {bad['g1']}, eager 2 parts {bad['e2']}, graphs 2 parts {bad['g2']}")
rep("last graphs 2 parts", t2.poses_out)
