"""Mints tests/golden/pipeline_golden_wide.npz: the REFERENCE's own hot-path Python (through tests/golden/ref_harness.py) on
MORE of the hypothesis set than pipeline_golden.npz (3 poses, fp32 only):

  w_*      32 poses = every 8th of the 252-pose grid, four of them moved so that their crop window leaves the frame
           (zero padding of the warp, clipped render), two at other depths:
             predict_pose_refine.make_crop_data_batch / predict_score.make_crop_data_batch network inputs (A, B; rows
             1::4, columns 2::4 are stored), PoseRefinePredictor.predict (1 iteration, amp off) refined poses + raw head
             outputs, ScorePredictor.predict (amp off) scores
  pair_*   exactly two poses through the refiner's make_crop_data_batch: the N == 2 broadcasting quirk of
           predict_pose_refine.py:44-45 (both hypotheses rendered with [umin_0, vmin_0, umax_1, vmax_1])
  un_*     use_normal=True through the refiner's make_crop_data_batch (3 poses): BatchPoseData.normalAs / normalBs -- the rendered
           normals and a synthetic frame normal map, both through the nearest warp by tf_to_crops -- which the reference
           stores and never feeds to the network
  amp_*    the first 8 of the 32 poses through both predictors with amp=True.  `torch.cuda.amp.autocast` needs a GPU; in
           the build container it is redirected to torch.autocast('cpu', dtype=float16): same cast policy for these
           modules except where the conv bias is added (CPU: fp32 accumulator; CUDA/ROCm: rounded fp16 output) -- see
           make_golden_amp.py.  oracle.nets_amp with CONV_BIAS='fused' restates exactly this configuration.

    python tests/golden/make_golden_pipeline_wide.py        (build container only; ~3 minutes)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

ROWS, COLS = slice(1, None, 4), slice(2, None, 4)
N_AMP = 8


def wide_poses(grid):
    """32 poses of the 252 grid, some moved (deterministic)"""
    P = grid[::8][:32].copy()
    P[3, :3, 3] = [0.17, 0.12, 0.6]        # window leaves the frame bottom right
    P[9, :3, 3] = [-0.20, -0.14, 0.62]     # ... top left
    P[17, :3, 3] = [0.0, 0.16, 0.75]       # ... bottom edge only
    P[25, :3, 3] = [-0.21, 0.0, 0.7]       # ... left edge only
    P[5, :3, 3] += [0.0, 0.0, 0.35]        # farther: window smaller than 160 px (magnifying warp)
    P[12, :3, 3] += [0.01, -0.01, -0.3]    # nearer: window larger than the object crop (minifying warp)
    return P.astype(np.float32)


def normal_map_for_tests(H=480, W=640):
    """a seeded (H,W,3) unit-normal map of the frame (any smooth field will do: it is only resampled)"""
    v, u = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    n = np.stack([np.sin(u * 0.031) * 0.6, np.cos(v * 0.043) * 0.5, -np.ones_like(u)], -1)
    return (n / np.linalg.norm(n, axis=-1, keepdims=True)).astype(np.float32)


def main():
    torch.set_num_threads(8)
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from make_golden_pipeline import Cfg, scene_dict
    from oracle import ops as oo
    from oracle import pipeline as op
    sc = scene_dict()
    import ref_harness as rh
    ns = rh.load_reference()
    out = {}
    K, diam, mesh = sc["K"], np.float64(sc["diameter"]), sc["mesh"]
    d = op.preprocess_depth(sc["depth"])
    xyz_map = oo.depth2xyzmap(d, K, f64_internal=True)
    mnp = op.mesh_tensors_np(mesh)
    mt = {"pos": torch.as_tensor(np.asarray(mesh.vertices), dtype=torch.float), "faces": torch.as_tensor(np.asarray(mesh.faces), dtype=torch.int),
          "vnormals": torch.as_tensor(np.asarray(mesh.vertex_normals), dtype=torch.float), "tex": torch.as_tensor(mnp["tex"])[None],
          "uv": torch.as_tensor(mnp["uv"]), "uv_idx": torch.as_tensor(mnp["uv_idx"])}
    P = wide_poses(sc["poses"])
    out["w_poses_in"] = P
    rcfg = Cfg(dict(DEFAULT_REFINE_CFG, input_resize=(160, 160)))   # tuple: see make_golden_pipeline.py
    scfg = Cfg(dict(DEFAULT_SCORE_CFG, input_resize=(160, 160)))
    rds = ns.h5_dataset.PoseRefinePairH5Dataset(cfg=rcfg, h5_file="", mode="test")
    sds = ns.h5_dataset.ScoreMultiPairH5Dataset(cfg=scfg, mode="test", h5_file=None, max_num_key=1)
    rgb_t, depth_t, xyz_t = torch.as_tensor(sc["rgb"], dtype=torch.float), torch.as_tensor(d), torch.as_tensor(xyz_map)
    t0 = time.time()
    pd = ns.refine.make_crop_data_batch(rcfg.input_resize, torch.as_tensor(P), mesh, rgb_t, depth_t, K, crop_ratio=rcfg["crop_ratio"],
                                        xyz_map=xyz_t, cfg=rcfg, glctx=None, mesh_tensors=mt, dataset=rds, mesh_diameter=diam)
    out["w_refine_A"] = torch.cat([pd.rgbAs, pd.xyz_mapAs], 1).numpy()[:, :, ROWS, COLS].astype(np.float32)
    out["w_refine_B"] = torch.cat([pd.rgbBs, pd.xyz_mapBs], 1).numpy()[:, :, ROWS, COLS].astype(np.float32)
    pd = ns.score.make_crop_data_batch(scfg.input_resize, torch.as_tensor(P), mesh, rgb_t, depth_t, K, crop_ratio=scfg["crop_ratio"],
                                       glctx=None, mesh_tensors=mt, dataset=sds, cfg=scfg, mesh_diameter=diam)
    out["w_score_A"] = torch.cat([pd.rgbAs, pd.xyz_mapAs], 1).numpy()[:, :, ROWS, COLS].astype(np.float32)
    out["w_score_B"] = torch.cat([pd.rgbBs, pd.xyz_mapBs], 1).numpy()[:, :, ROWS, COLS].astype(np.float32)
    print(f"inputs: {time.time() - t0:.1f} s")
    # ---- N == 2 quirk
    P2 = P[[5, 20]].copy()        # different depths: two different windows, so the mixed bbox is visible
    pd = ns.refine.make_crop_data_batch(rcfg.input_resize, torch.as_tensor(P2), mesh, rgb_t, depth_t, K, crop_ratio=rcfg["crop_ratio"],
                                        xyz_map=xyz_t, cfg=rcfg, glctx=None, mesh_tensors=mt, dataset=rds, mesh_diameter=diam)
    out["pair_poses_in"] = P2
    out["pair_refine_A"] = torch.cat([pd.rgbAs, pd.xyz_mapAs], 1).numpy()[:, :, ::2, ::2].astype(np.float32)
    out["pair_refine_B"] = torch.cat([pd.rgbBs, pd.xyz_mapBs], 1).numpy()[:, :, ::2, ::2].astype(np.float32)
    # ---- use_normal=True (predict_pose_refine.py:50,58,75-76): normalAs / normalBs of the batch (never fed to the network)
    ncfg = Cfg(dict(rcfg, use_normal=True))
    nds = ns.h5_dataset.PoseRefinePairH5Dataset(cfg=ncfg, h5_file="", mode="test")
    nm = normal_map_for_tests()
    P3 = P[[0, 3, 12]].copy()
    pd = ns.refine.make_crop_data_batch(ncfg.input_resize, torch.as_tensor(P3), mesh, rgb_t, depth_t, K, crop_ratio=ncfg["crop_ratio"],
                                        xyz_map=xyz_t, normal_map=nm, cfg=ncfg, glctx=None, mesh_tensors=mt, dataset=nds, mesh_diameter=diam)
    out["un_poses_in"] = P3
    out["un_normalAs"] = pd.normalAs.numpy()[:, :, ::2, ::2].astype(np.float32)
    out["un_normalBs"] = pd.normalBs.numpy()[:, :, ::2, ::2].astype(np.float32)
    out["un_A"] = torch.cat([pd.rgbAs, pd.xyz_mapAs], 1).numpy()[:, :, ::4, ::4].astype(np.float32)   # unchanged by the flag
    # ---- predictors, fp32
    rsd, ssd = random_state_dict("refine", dict(rcfg), 0), random_state_dict("score", dict(scfg), 0)

    def refiner(amp):
        rp = object.__new__(ns.refine.PoseRefinePredictor)
        rp.amp, rp.cfg, rp.dataset = amp, rcfg, rds
        rp.model = ns.refine_network.RefineNet(cfg=rcfg, c_in=6).eval()
        rp.model.load_state_dict(rsd, strict=True)
        rp.last_trans_update = rp.last_rot_update = None
        return rp

    def scorer(amp):
        sp = object.__new__(ns.score.ScorePredictor)
        sp.amp, sp.cfg, sp.dataset = amp, scfg, sds
        sp.model = ns.score_network.ScoreNetMultiPair(cfg=scfg, c_in=6).eval()
        sp.model.load_state_dict(ssd, strict=True)
        return sp

    t0 = time.time()
    rp = refiner(False)
    cap = {}
    hk = rp.model.register_forward_hook(lambda m, i, o: cap.update(trans=o["trans"].detach().clone(), rot=o["rot"].detach().clone()))
    refined, _ = rp.predict(sc["rgb"], d, K, P, xyz_map, mesh=mesh, mesh_tensors=mt, glctx=None, mesh_diameter=diam, iteration=1)
    hk.remove()
    out["w_refined_1it"] = refined.numpy().astype(np.float32)
    out["w_raw_trans"], out["w_raw_rot"] = cap["trans"].numpy().astype(np.float32), cap["rot"].numpy().astype(np.float32)
    scores, _ = scorer(False).predict(sc["rgb"], d, K, P, mesh=mesh, mesh_tensors=mt, glctx=None, mesh_diameter=diam)
    out["w_scores"] = scores.numpy().astype(np.float32)
    refined2, _ = rp.predict(sc["rgb"], d, K, P2, xyz_map, mesh=mesh, mesh_tensors=mt, glctx=None, mesh_diameter=diam, iteration=1)
    out["pair_refined_1it"] = refined2.numpy().astype(np.float32)
    print(f"fp32 predictors: {time.time() - t0:.1f} s")
    # ---- predictors, amp=True (CPU autocast stands in for torch.cuda.amp.autocast)
    t0 = time.time()
    real = torch.cuda.amp.autocast
    torch.cuda.amp.autocast = lambda enabled=True, **kw: torch.autocast("cpu", dtype=torch.float16, enabled=enabled)
    try:
        rp = refiner(True)
        hk = rp.model.register_forward_hook(lambda m, i, o: cap.update(trans=o["trans"].detach().clone(), rot=o["rot"].detach().clone()))
        refined, _ = rp.predict(sc["rgb"], d, K, P[:N_AMP], xyz_map, mesh=mesh, mesh_tensors=mt, glctx=None, mesh_diameter=diam, iteration=1)
        hk.remove()
        assert cap["trans"].dtype == torch.float16, cap["trans"].dtype      # autocast was really on
        out["amp_refined_1it"] = refined.numpy().astype(np.float32)
        out["amp_raw_trans"], out["amp_raw_rot"] = cap["trans"].float().numpy(), cap["rot"].float().numpy()
        scores, _ = scorer(True).predict(sc["rgb"], d, K, P[:N_AMP], mesh=mesh, mesh_tensors=mt, glctx=None, mesh_diameter=diam)
        out["amp_scores"] = scores.float().numpy().astype(np.float32)
    finally:
        torch.cuda.amp.autocast = real
    print(f"amp predictors: {time.time() - t0:.1f} s")
    path = os.path.join(HERE, "pipeline_golden_wide.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k, v in out.items():
        print(f"  {k}: {v.shape} {v.dtype}")


if __name__ == "__main__":
    main()
