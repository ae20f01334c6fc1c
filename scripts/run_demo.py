#!/usr/bin/env python
"""Demo driver with the reference's call sequence and arguments (run_demo.py:15-78): load a mesh, register on the
first frame, track the following frames, write ob_in_cam/<frame>.txt.  `--synthetic N` first writes an N-frame
synthetic sequence (the textured can moving in front of the camera) in the demo layout, because the reference's
demo_data and weights are not redistributable; with real data pass --mesh_file / --test_scene_dir and put the
checkpoints under $FOUNDATIONPOSE_WEIGHTS.  Visualisation (debug >= 1 overlays) is not implemented."""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_synthetic_demo(out_dir, n_frames, dev):
    """textured can, smooth motion, rendered with the product rasteriser; returns (mesh_file, scene_dir, gt poses)"""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.datareader import write_sequence
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.mesh_io import save_obj
    from foundationpose_amd.Utils import euler_matrix, make_mesh_tensors, nvdiffrast_render
    mesh = make_can_mesh()
    os.makedirs(os.path.join(out_dir, "mesh"), exist_ok=True)
    mesh_file = os.path.join(out_dir, "mesh", "textured_simple.obj")
    save_obj(mesh, mesh_file)
    gm = make_mesh_tensors(mesh, device=dev)
    T0 = syn.gt_pose(0)
    poses = []
    for i in range(n_frames):
        T = T0.copy()
        T[:3, :3] = T0[:3, :3] @ euler_matrix(0.02 * i, 0.015 * i, 0.0)[:3, :3]
        T[:3, 3] = T0[:3, 3] + np.array([0.002 * i, -0.001 * i, 0.003 * i])
        poses.append(T)
    color, depth, _ = nvdiffrast_render(K=syn.YCBV_K, H=syn.H, W=syn.W, ob_in_cams=torch.as_tensor(np.stack(poses), device=dev, dtype=torch.float),
                                        mesh_tensors=gm, use_light=True, extra={})
    cs, ds, ms = [], [], []
    for i in range(n_frames):
        rgb, d, mask = syn.compose_frame(color[i].cpu().numpy(), depth[i].cpu().numpy(), seed=i)
        cs.append(rgb); ds.append(d); ms.append(mask)
    write_sequence(out_dir, syn.YCBV_K, cs, ds, ms, gt_poses=poses)
    return mesh_file, out_dir, poses


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh_file", type=str, default=None)
    ap.add_argument("--test_scene_dir", type=str, default=None)
    ap.add_argument("--est_refine_iter", type=int, default=5)
    ap.add_argument("--track_refine_iter", type=int, default=2)
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--debug_dir", type=str, default=os.path.join(ROOT, "gpurun_out", "demo_debug"))
    ap.add_argument("--synthetic", type=int, default=0, help="write an N-frame synthetic demo sequence first and run on it")
    ap.add_argument("--track_graph", action="store_true", help="replay track_one as one captured hipGraph per frame")
    ap.add_argument("--standin_weights", action="store_true", help="seeded stand-in checkpoints instead of weights/")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="[%(funcName)s()] %(message)s")

    from foundationpose_amd import dr
    from foundationpose_amd.datareader import YcbineoatReader
    from foundationpose_amd.estimater import FoundationPose
    from foundationpose_amd.mesh_io import load_mesh
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.Utils import set_seed
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict

    set_seed(0)
    dev = torch.device("cuda:0")
    gt = None
    if args.synthetic > 0:
        args.mesh_file, args.test_scene_dir, gt = write_synthetic_demo(os.path.join(args.debug_dir, "synthetic_scene"), args.synthetic, dev)
        args.standin_weights = True
    if not args.mesh_file or not args.test_scene_dir:
        ap.error("--mesh_file and --test_scene_dir are required (or --synthetic N)")
    mesh = load_mesh(args.mesh_file)
    os.makedirs(os.path.join(args.debug_dir, "ob_in_cam"), exist_ok=True)
    if args.standin_weights:
        scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), device=dev)
        refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
    else:
        scorer, refiner = ScorePredictor(device=dev), PoseRefinePredictor(device=dev)
    glctx = dr.RasterizeCudaContext()
    est = FoundationPose(model_pts=mesh.vertices, model_normals=mesh.vertex_normals, mesh=mesh, scorer=scorer, refiner=refiner,
                         debug_dir=args.debug_dir, debug=args.debug, glctx=glctx, device=dev, track_graph=args.track_graph)
    logging.info("estimator initialization done")
    reader = YcbineoatReader(video_dir=args.test_scene_dir, shorter_side=None, zfar=np.inf)
    times = []
    for i in range(len(reader.color_files)):
        color, depth = reader.get_color(i), reader.get_depth(i)
        t0 = time.perf_counter()
        if i == 0:
            mask = reader.get_mask(0).astype(bool)
            pose = est.register(K=reader.K, rgb=color, depth=depth, ob_mask=mask, iteration=args.est_refine_iter)
        else:
            pose = est.track_one(rgb=color, depth=depth, K=reader.K, iteration=args.track_refine_iter)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        np.savetxt(os.path.join(args.debug_dir, "ob_in_cam", f"{reader.id_strs[i]}.txt"), pose.reshape(4, 4))
    logging.info(f"register {times[0] * 1e3:.1f} ms; track_one median {np.median(times[1:]) * 1e3 if len(times) > 1 else float('nan'):.2f} ms/frame "
                 f"over {len(times) - 1} frames; poses in {args.debug_dir}/ob_in_cam")
    return times


if __name__ == "__main__":
    main()
