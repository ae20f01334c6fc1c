"""phase timers of k_conv_sw over whole STEPS of the bench workload (profiling build): cold start / main loop / epilogue per tile, summed
over every launch of the step, with the sub-batches on ONE stream and on concurrent streams -- is the ~8 us epilogue of a tile (measured
with one layer alone on the chip, every CU reaching its epilogue in the same microsecond) also what a tile pays in the mode that is timed?
    FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so python scripts/dbg_conv_sw_step.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from foundationpose_amd import ops
from foundationpose_amd.overlap import reserve_streams
from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
from foundationpose_amd.predict_score import ScorePredictor
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict

dev = torch.device("cuda:0")
reserve_streams(dev, 1)
L = C.CDLL(os.environ["FP_AMD_LIB"])
sc = bench.build_scene(dev, 0, 252)
opts = dict(device=dev, precision="fp16", n_streams=2, graph=False)
refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), **opts)
scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), **opts)
rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
depth_t = ops.bilateral_filter_depth(ops.erode_depth(torch.as_tensor(sc["depth"], device=dev)))
xyz_t = ops.depth_to_xyz(depth_t, sc["K"], f64_internal=True)
poses0 = torch.as_tensor(sc["poses"], device=dev)


def step():
    p, _ = refiner.predict(rgb_t, depth_t, sc["K"], poses0, xyz_t, mesh=sc["mesh"], mesh_tensors=sc["gm"], mesh_diameter=sc["diameter"], iteration=5)
    s, _ = scorer.predict(rgb_t, depth_t, sc["K"], p, mesh=sc["mesh"], mesh_tensors=sc["gm"], mesh_diameter=sc["diameter"])
    return s


res = {}
for mode in ("serialized", "concurrent", "serialized", "concurrent"):
    refiner.sub.serial = scorer.sub.serial = mode == "serialized"
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    L.fp_dbg_conv_sw(out, 1)
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    L.fp_dbg_conv_sw(out, 0)
    n = out[3]
    r = dict(ms_per_step=dt, tiles_per_step=n / K, cold_us=out[0] / n / 100, loop_us=out[1] / n / 100, epilogue_us=out[2] / n / 100,
             cu_ms_per_step=(out[0] + out[1] + out[2]) / 100 / 1e3 / K / 256)
    print(mode, json.dumps(r), flush=True)
