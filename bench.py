#!/usr/bin/env python
"""Headline benchmark: pose-hypotheses/sec (raster+refine+score), 252 hyp x 160x160 @ 640x480 RGB-D.

A step = one pass of the hot path for one object and one frame: R=5 refine iterations (crop windows -> fused
rasteriser -> observed crop -> RefineNet -> pose update) + one score pass (-> ScoreNet -> ranking), inputs resident
in HBM (BASELINE.json configs[1]).  With --gpus N every rank owns one object (configs[3]) and the
per-object {scores, refined poses} records are exchanged with ONE RCCL all-gather per step; per-GPU work is fixed
(weak scaling).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA
REFINE_GFLOP_PER_HYP = 23.946
SCORE_GFLOP_PER_HYP = 21.938
SCORE_GFLOP_CROSS_252 = 0.659


def build_scene(dev, seed, n_hyp):
    """one synthetic object + frame on the device; frame rendered with the product's own rasteriser"""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.Utils import euler_matrix, make_mesh_tensors, nvdiffrast_render, sample_views_icosphere
    from foundationpose_amd.mesh import make_can_mesh
    mesh = make_can_mesh(seed=seed)
    gm = make_mesh_tensors(mesh, device=dev)
    K = syn.YCBV_K.copy()
    T = syn.gt_pose(seed)
    color, depth, _ = nvdiffrast_render(K=K, H=syn.H, W=syn.W, ob_in_cams=torch.as_tensor(T[None], device=dev, dtype=torch.float),
                                        mesh_tensors=gm, use_light=True, extra={})
    rgb, d, mask = syn.compose_frame(color[0].cpu().numpy(), depth[0].cpu().numpy(), seed=seed)
    cams = sample_views_icosphere(40)
    grid = np.asarray([np.linalg.inv(c @ euler_matrix(0, 0, a)) for c in cams for a in np.deg2rad(np.arange(0, 360, 60))])
    grid[:, :3, 3] = T[:3, 3] + np.array([0.004, -0.003, 0.01])
    reps = int(np.ceil(n_hyp / len(grid)))
    grid = np.tile(grid, (reps, 1, 1))[:n_hyp]
    diameter = float(np.linalg.norm(mesh.vertices.max(0) - mesh.vertices.min(0)))
    return dict(mesh=mesh, gm=gm, K=K, rgb=rgb, depth=d, mask=mask, poses=grid.astype(np.float32), diameter=diameter, T=T)


def cpu_baseline(n_hyp=16, iters=5):
    """oracle (C rasteriser/warp with OpenMP + torch-CPU fp32 networks) on BASELINE configs[0]; checker, not product"""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.Utils import euler_matrix, sample_views_icosphere
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from oracle import ops as oo
    from oracle import pipeline as op
    mesh = make_can_mesh()
    mnp = op.mesh_tensors_np(mesh)
    K, T = syn.YCBV_K, syn.gt_pose(0)
    full = oo.render_crops(mnp, T[None].astype(np.float32), None, K, syn.H, syn.W, (syn.H, syn.W), normalize_xyz=False,
                           want=("color", "depth"))
    rgb, depth, _ = syn.compose_frame(full["color"][0], full["depth"][0])
    cams = sample_views_icosphere(40)
    grid = np.asarray([np.linalg.inv(c @ euler_matrix(0, 0, a)) for c in cams for a in np.deg2rad(np.arange(0, 360, 60))])
    grid[:, :3, 3] = T[:3, 3] + np.array([0.004, -0.003, 0.01])
    poses = grid[:n_hyp].astype(np.float32)
    diam = float(np.linalg.norm(mesh.vertices.max(0) - mesh.vertices.min(0)))
    rcfg, scfg = dict(DEFAULT_REFINE_CFG), dict(DEFAULT_SCORE_CFG)
    rsd, ssd = random_state_dict("refine", rcfg, 0), random_state_dict("score", scfg, 0)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    d = op.preprocess_depth(depth)
    xyz = oo.depth2xyzmap(d, K)
    op.refine_predict(rcfg, rsd, rgb, d, K, poses[:2], xyz, mnp, diam, iteration=1)  # warm-up
    t0 = time.perf_counter()
    p = op.refine_predict(rcfg, rsd, rgb, d, K, poses, xyz, mnp, diam, iteration=iters)
    s = op.score_predict(scfg, ssd, rgb, d, K, p, mnp, diam)
    np.argsort(-s)
    dt = time.perf_counter() - t0
    return dict(value=n_hyp / dt, unit="pose-hypotheses/sec", cores=int(cores), kind="port",
                sample=f"{n_hyp} hypotheses x ({iters} refine + 1 score), one frame, oracle C raster/warp (OpenMP {oo.num_threads()} thr) "
                       f"+ torch-CPU fp32 nets ({torch.get_num_threads()} thr), {dt:.2f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hyps", type=int, default=252)
    ap.add_argument("--refine-iters", type=int, default=5)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nchw", action="store_true", help="disable channels_last activations")
    ap.add_argument("--no-hip-gemm", action="store_true", help="route conv1/QKV through PyTorch instead of the MFMA kernels")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from foundationpose_amd import ops
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict

    N, R = args.hyps, args.refine_iters
    sc = build_scene(dev, seed=rank, n_hyp=N)
    opts = dict(device=dev, precision=args.precision, channels_last=not args.nchw, use_hip_gemm=not args.no_hip_gemm)
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), **opts)
    scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), **opts)
    rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
    depth_t = ops.bilateral_filter_depth(ops.erode_depth(torch.as_tensor(sc["depth"], device=dev)))
    xyz_t = ops.depth_to_xyz(depth_t, sc["K"], f64_internal=True)
    poses0 = torch.as_tensor(sc["poses"], device=dev)
    gather_buf = [torch.empty((N, 17), device=dev) for _ in range(world)] if world > 1 else None

    def step():
        p, _ = refiner.predict(rgb_t, depth_t, sc["K"], poses0, xyz_t, mesh=sc["mesh"], mesh_tensors=sc["gm"],
                               mesh_diameter=sc["diameter"], iteration=R)
        s, _ = scorer.predict(rgb_t, depth_t, sc["K"], p, mesh=sc["mesh"], mesh_tensors=sc["gm"],
                              mesh_diameter=sc["diameter"])
        ids = s.argsort(descending=True)
        rec = torch.cat([s[ids, None], p[ids].reshape(N, 16)], dim=1)  # fused record: scores || refined poses
        if world > 1:
            dist.all_gather(gather_buf, rec)   # ONE RCCL all-gather per register()
        return rec

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    timers = ops.KernelTimers()
    t0 = time.perf_counter()
    with timers:
        for _ in range(args.steps):
            rec = step()
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert torch.isfinite(rec).all()

    if rank == 0:
        ksum = timers.summary()
        V, T = sc["gm"]["_handle"].V, sc["gm"]["_handle"].T
        esz = 2 if args.precision == "fp16" else 4
        # algorithmic bytes per launch: A output + mesh read once per hypothesis
        render_bytes = N * (6 * 160 * 160 * esz + 32 * V + 12 * T)
        warp_bytes = N * 6 * 160 * 160 * esz + 480 * 640 * 24
        stage_bytes = render_bytes + warp_bytes
        kern = {}
        for name, (calls, ms) in ksum.items():
            kern[name] = dict(calls=calls, avg_ms=round(ms, 5))
        r_ms = ksum.get("fp_render_crops", (0, float("nan")))[1]
        w_ms = ksum.get("fp_warp_crops", (0, float("nan")))[1]
        ach = render_bytes / (r_ms * 1e-3) / 1e9
        flops = world * N * (R * REFINE_GFLOP_PER_HYP + SCORE_GFLOP_PER_HYP) + world * SCORE_GFLOP_CROSS_252 * (N / 252.0) ** 2
        out = {
            "metric": "pose-hypotheses/sec (raster+refine+score), 252 hyp x 160x160 @ 640x480 RGB-D",
            "value": world * N * args.steps / dt, "unit": "pose-hypotheses/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.precision == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: synthetic can (V={V}, T={T}), one 640x480 RGB-D frame per rank, "
                                   f"{N} hypotheses, {R} refine iterations + 1 score pass, 160x160 crops, random-init weights",
                       "hypotheses_per_gpu": N, "refine_iterations": R, "parallelism": f"object-parallel x{world}, one all-gather/step"},
            "roofline": {"kernel": "k_render (fp_render_crops: fused vertex+raster+shade+normalise+concat)", "bound": "hbm",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_launch": render_bytes, "avg_launch_ms": r_ms},
            "stage_raster_crop": {"bytes_per_pass": stage_bytes, "ms_per_pass": r_ms + w_ms,
                                  "achieved_GBps": stage_bytes / ((r_ms + w_ms) * 1e-3) / 1e9,
                                  "frac_of_hbm_peak": stage_bytes / ((r_ms + w_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "network_mfma": {"algorithmic_TFLOP_per_step": flops / 1e3, "achieved_TFLOPs": flops / 1e3 / (dt / args.steps),
                             "frac_of_mfma_peak": flops / 1e3 / (dt / args.steps) / (MFMA_PEAK_TFLOPS * world)},
            "kernels": kern,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
