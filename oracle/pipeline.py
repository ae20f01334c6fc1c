"""CPU restatement of the predictor / estimator control flow.  TEST INFRASTRUCTURE ONLY.

Follows predict_pose_refine.py:26-89,150-239, predict_score.py:56-114,161-226 and estimater.py:159-268 with the image
ops of oracle.ops (C) and the networks of oracle.nets (torch CPU fp32)."""
import numpy as np
import torch

from . import nets, nets_amp, ops


def _nets(amp):
    """amp=False: fp32 networks (oracle/nets.py); amp=True: the reference's deployed fp16-autocast policy (oracle/nets_amp.py)"""
    return nets_amp if amp else nets


def mesh_tensors_np(mesh):
    """numpy equivalent of make_mesh_tensors (Utils.py:104-130)."""
    t = {"pos": np.asarray(mesh.vertices, dtype=np.float32), "faces": np.asarray(mesh.faces, dtype=np.int32),
         "vnormals": np.asarray(mesh.vertex_normals, dtype=np.float32)}
    visual = getattr(mesh, "visual", None)
    material = getattr(visual, "material", None)
    image = getattr(material, "image", None) if material is not None else None
    if image is not None and getattr(visual, "uv", None) is not None:
        img = np.asarray(image)[..., :3]
        t["tex"] = img.astype(np.float32) * (np.float32(1.0) / np.float32(255.0))  # torch GPU `/255.0` = mul by f32 reciprocal
        uv = np.asarray(visual.uv, dtype=np.float32).copy()
        uv[:, 1] = 1 - uv[:, 1]
        t["uv"] = uv
        t["uv_idx"] = t["faces"]
    else:
        vc = visual.vertex_colors if (visual is not None and visual.vertex_colors is not None) else \
            np.tile(np.array([128, 128, 128]).reshape(1, 3), (len(mesh.vertices), 1))
        t["vertex_color"] = np.asarray(vc)[..., :3].astype(np.float32) * (np.float32(1.0) / np.float32(255.0))
    return t


def refine_inputs(cfg, poses, mesh_np, rgb, xyz_map, K, mesh_diameter):
    """-> A, B (N,6,h,w) f32, tf_to_crops, bbox2d   (make_crop_data_batch + transform_batch + concat)"""
    H, W = rgb.shape[:2]
    oh, ow = cfg["input_resize"]
    poses = np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4)
    tf, bb = ops.crop_windows(poses, K, mesh_diameter, cfg["crop_ratio"], (ow, oh))
    if poses.shape[0] == 2:  # reference broadcasting quirk, SURVEY App. D.5
        bb = np.tile(np.array([bb[0, 0], bb[0, 1], bb[1, 2], bb[1, 3]], np.float32)[None], (2, 1))
    A = ops.render_crops(mesh_np, poses, bb, K, H, W, (oh, ow), mesh_diameter, 0.001, bool(cfg["normalize_xyz"]),
                         want=("A",))["A"]
    B = ops.warp_crops(np.asarray(rgb, dtype=np.float32), xyz_map, None, tf, K, poses, mesh_diameter, ops.MODE_REFINE,
                       bool(cfg["normalize_xyz"]), (oh, ow))
    return A, B, tf, bb


def score_inputs(cfg, poses, mesh_np, rgb, depth, K, mesh_diameter):
    H, W = rgb.shape[:2]
    oh, ow = cfg["input_resize"]
    poses = np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4)
    tf, bb = ops.crop_windows(poses, K, mesh_diameter, cfg["crop_ratio"], (ow, oh))
    A = ops.render_crops(mesh_np, poses, bb, K, H, W, (oh, ow), mesh_diameter, 0.1, bool(cfg["normalize_xyz"]),
                         want=("A",))["A"]
    B = ops.warp_crops(np.asarray(rgb, dtype=np.float32), None, depth, tf, K, poses, mesh_diameter, ops.MODE_SCORE,
                       bool(cfg["normalize_xyz"]), (oh, ow))
    return A, B, tf, bb


def refine_predict(cfg, sd, rgb, depth, K, ob_in_cams, xyz_map, mesh_np, mesh_diameter, iteration=5, trace=None, amp=False):
    poses = np.asarray(ob_in_cams, dtype=np.float32).reshape(-1, 4, 4).copy()
    tn = cfg["trans_normalizer"]
    tn = [float(tn)] * 3 if isinstance(tn, (int, float)) else [float(v) for v in tn]
    for it in range(iteration):
        A, B, tf, _ = refine_inputs(cfg, poses, mesh_np, rgb, xyz_map, K, mesh_diameter)
        out = _nets(amp).refine_forward(torch.from_numpy(A), torch.from_numpy(B), sd)
        if cfg.get("trans_rep", "tracknet") == "deepim":   # predict_pose_refine.py:201-215
            dt = ops.deepim_trans_delta(out["trans"].numpy(), poses, K, tf, cfg["input_resize"][0], bool(cfg["normalize_xyz"]), mesh_diameter)
            # the C pose update with an identity translation scale: normalize_xyz=True multiplies by diameter/2 = 1
            poses = ops.pose_update(dt, out["rot"].numpy(), poses, cfg["rot_rep"], True, tn, float(cfg["rot_normalizer"]), 2.0)
        elif cfg.get("trans_rep", "tracknet") != "tracknet" and not cfg["normalize_xyz"]:   # plain else, predict_pose_refine.py:217-218
            # raw output as the translation: the C pose update with normalize_xyz=True and diameter 2 multiplies by 1
            poses = ops.pose_update(out["trans"].numpy(), out["rot"].numpy(), poses, cfg["rot_rep"], True, tn, float(cfg["rot_normalizer"]), 2.0)
        else:
            poses = ops.pose_update(out["trans"].numpy(), out["rot"].numpy(), poses, cfg["rot_rep"],
                                    bool(cfg["normalize_xyz"]), tn, float(cfg["rot_normalizer"]), float(mesh_diameter))
        if trace is not None:
            trace.append(dict(A=A, B=B, trans=out["trans"].numpy(), rot=out["rot"].numpy(), poses=poses.copy()))
    return poses


def score_predict(cfg, sd, rgb, depth, K, ob_in_cams, mesh_np, mesh_diameter, trace=None, amp=False):
    A, B, _, _ = score_inputs(cfg, ob_in_cams, mesh_np, rgb, depth, K, mesh_diameter)
    N = A.shape[0]
    out = _nets(amp).score_forward(torch.from_numpy(A), torch.from_numpy(B), sd, L=N)
    if trace is not None:
        trace.append(dict(A=A, B=B))
    return out["score_logit"].reshape(-1).numpy() + 100.0


def preprocess_depth(depth):
    return ops.bilateral_filter_depth(ops.erode_depth(depth, radius=2), radius=2)


def register(refine_cfg, refine_sd, score_cfg, score_sd, K, rgb, depth, poses0, mesh_np, mesh_diameter, iteration=5, amp=False):
    """estimater.py:159-240 after hypothesis generation: -> (sorted poses, sorted scores, order)."""
    d = preprocess_depth(depth)
    xyz_map = ops.depth2xyzmap(d, K, f64_internal=True)
    poses = refine_predict(refine_cfg, refine_sd, rgb, d, K, poses0, xyz_map, mesh_np, mesh_diameter, iteration, amp=amp)
    scores = score_predict(score_cfg, score_sd, rgb, d, K, poses, mesh_np, mesh_diameter, amp=amp)
    order = np.argsort(-scores, kind="stable")
    return poses[order], scores[order], order
