#!/usr/bin/env python
"""Benchmark driver with the reference's call sequence (run_ycb_video.py:43-130): per object `reset_object`, per
keyframe `register(iteration=est_refine_iter)` with the instance's visible mask, results as
`{video_id: {frame id: {ob_id: 4x4}}}` in `<debug_dir>/ycbv_res.yml` -- plus what the reference leaves to an external
script: ADD / ADD-S per estimate and their AUC (Utils.py:232-266).  `--synthetic N` first mints an N-frame scene in the
BOP layout from the synthetic can (datasets and released weights are not in this container); real data:
`--ycbv_dir <BOP ycbv root>` with `$YCB_VIDEO_DIR/models` or `--models_dir`."""
import argparse
import glob
import logging
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_synthetic_bop(root, n_frames, dev, ob_id=1):
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.datareader import write_bop_scene
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.Utils import euler_matrix, make_mesh_tensors, nvdiffrast_render
    mesh = make_can_mesh()
    gm = make_mesh_tensors(mesh, device=dev)
    T0 = syn.gt_pose(0)
    poses = []
    for i in range(n_frames):
        T = T0.copy()
        T[:3, :3] = T0[:3, :3] @ euler_matrix(0.3 * i, 0.2 * i, 0.1 * i)[:3, :3]
        T[:3, 3] = T0[:3, 3] + np.array([0.02 * i, -0.01 * i, 0.03 * i])
        poses.append(T)
    color, depth, _ = nvdiffrast_render(K=syn.YCBV_K, H=syn.H, W=syn.W, ob_in_cams=torch.as_tensor(np.stack(poses), device=dev, dtype=torch.float),
                                        mesh_tensors=gm, use_light=True, extra={})
    cs, ds, inst = [], [], []
    for i in range(n_frames):
        rgb, d, mask = syn.compose_frame(color[i].cpu().numpy(), depth[i].cpu().numpy(), seed=i)
        cs.append(rgb); ds.append(d); inst.append([(ob_id, poses[i], mask)])
    write_bop_scene(os.path.join(root, "test", "000001"), syn.YCBV_K, cs, ds, inst, models_dir=os.path.join(root, "models"), meshes={ob_id: mesh})
    return root


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ycbv_dir", type=str, default=None, help="BOP dataset root with test/<scene>/")
    ap.add_argument("--models_dir", type=str, default=None)
    ap.add_argument("--est_refine_iter", type=int, default=5)
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--debug_dir", type=str, default=os.path.join(ROOT, "gpurun_out", "ycbv_debug"))
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--standin_weights", action="store_true")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="[%(funcName)s()] %(message)s")

    from foundationpose_amd import dr, vis
    from foundationpose_amd.datareader import YcbVideoReader
    from foundationpose_amd.estimater import FoundationPose
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.Utils import set_seed
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict

    set_seed(0)
    dev = torch.device("cuda:0")
    os.makedirs(args.debug_dir, exist_ok=True)
    if args.synthetic > 0:
        args.ycbv_dir = write_synthetic_bop(os.path.join(args.debug_dir, "synthetic_bop"), args.synthetic, dev)
        args.models_dir = os.path.join(args.ycbv_dir, "models")
        args.standin_weights = True
    if not args.ycbv_dir:
        ap.error("--ycbv_dir is required (or --synthetic N)")
    if args.standin_weights:
        scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), device=dev)
        refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
    else:
        scorer, refiner = ScorePredictor(device=dev), PoseRefinePredictor(device=dev)
    video_dirs = sorted(glob.glob(f"{args.ycbv_dir}/test/*"))
    reader_tmp = YcbVideoReader(video_dirs[0], models_dir=args.models_dir)
    box = make_can_mesh(n_ang=8, n_axial=2, textured=False)        # placeholder object until reset_object (the reference uses a unit box)
    est = FoundationPose(model_pts=box.vertices.copy(), model_normals=box.vertex_normals.copy(), symmetry_tfs=None, mesh=box, scorer=scorer,
                         refiner=refiner, glctx=dr.RasterizeCudaContext(), debug_dir=args.debug_dir, debug=args.debug, device=dev)
    res, errs = {}, {"add": [], "adds": []}
    for ob_id in reader_tmp.ob_ids:
        mesh = reader_tmp.get_gt_mesh(ob_id)
        jobs = []
        for video_dir in video_dirs:
            reader = YcbVideoReader(video_dir, zfar=1.5, models_dir=args.models_dir)
            if ob_id not in reader.get_instance_ids_in_image(0):
                continue
            jobs += [(reader, i) for i in range(len(reader)) if reader.is_keyframe(i)]
        if not jobs:
            continue
        est.reset_object(model_pts=mesh.vertices.copy(), model_normals=mesh.vertex_normals.copy(), symmetry_tfs=reader_tmp.symmetry_tfs[ob_id], mesh=mesh)
        pts = np.asarray(mesh.vertices)
        for reader, i in jobs:
            mask = reader.get_mask(i, ob_id)
            if mask is None:
                continue
            pose = est.register(K=reader.get_K(i), rgb=reader.get_color(i), depth=reader.get_depth(i), ob_mask=mask, ob_id=ob_id, iteration=args.est_refine_iter)
            res.setdefault(reader.get_video_id(), {}).setdefault(reader.id_strs[i], {})[int(ob_id)] = np.asarray(pose).reshape(4, 4).tolist()
            gt = reader.get_gt_pose(i, ob_id, mask=mask)
            errs["add"].append(vis.add_err(pose, gt, pts))
            errs["adds"].append(vis.adds_err(pose, gt, pts))
    with open(os.path.join(args.debug_dir, "ycbv_res.yml"), "w") as f:
        yaml.safe_dump(res, f)
    summary = {"n": len(errs["add"]), "ADD_AUC": vis.compute_auc(errs["add"]) if errs["add"] else None,
               "ADDS_AUC": vis.compute_auc(errs["adds"]) if errs["adds"] else None,
               "ADD_mean_m": float(np.mean(errs["add"])) if errs["add"] else None, "ADDS_mean_m": float(np.mean(errs["adds"])) if errs["adds"] else None}
    logging.info(f"{summary}; poses in {args.debug_dir}/ycbv_res.yml")
    return summary


if __name__ == "__main__":
    main()
