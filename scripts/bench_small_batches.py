"""refine loop latency for SMALL hypothesis counts (the reference's track_one is N = 1; a tracker with a handful of hypotheses is N = 2..16):
predict(N, iteration=2), eager launches, with the encoder's convolutions on the plain entry point and on the split-K one
(engine.SPLITK_MAX_HYPS) -- where does splitting the k range stop paying?   python scripts/bench_small_batches.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from foundationpose_amd import engine, ops
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict

dev = torch.device("cuda:0")
sc = bench.build_scene(dev, 0, 252)
rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
depth_t = ops.bilateral_filter_depth(ops.erode_depth(torch.as_tensor(sc["depth"], device=dev)))
xyz_t = ops.depth_to_xyz(depth_t, sc["K"], f64_internal=True)
poses = torch.as_tensor(sc["poses"], device=dev)
rows = []
for n in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    res = {}
    for name, T in (("plain", 0), ("splitk", engine.SMALL_CALL_CEILING)):
        engine.SPLITK_MAX_HYPS = T            # a measuring script: no restore; the product path is engine.overrides()
        ref = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev, graph=False)
        run = lambda: ref.predict(rgb_t, depth_t, sc["K"], poses[:n], xyz_t, mesh=sc["mesh"], mesh_tensors=sc["gm"], mesh_diameter=sc["diameter"], iteration=2)
        for _ in range(3):
            out = run()[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
        res[name + "_out"] = out.clone()
    dmax = float((res["plain_out"] - res["splitk_out"]).abs().max())
    rows.append(dict(n=n, plain_ms=round(res["plain"], 3), splitk_ms=round(res["splitk"], 3), max_abs_pose_difference=dmax))
    print(json.dumps(rows[-1]), flush=True)
