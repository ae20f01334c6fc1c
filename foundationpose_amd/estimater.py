"""FoundationPose estimator facade -- drop-in for /root/reference/estimater.py:18-268
(``register`` / ``track_one`` / ``reset_object`` / ``to_device``), orchestrating the HIP hot path.

Differences that are deliberate and documented in DESIGN.md:
  * depth filtering, back-projection and the translation guess (masked median) stay on the device (the reference
    round-trips numpy<->GPU four times); register() moves a handful of scalars over PCIe, never the depth map;
  * open3d voxel down-sampling of ``self.pts``/``self.normals`` (computed but unused by the hot path)
    is replaced by the raw model points;
  * no global ``torch.set_default_tensor_type`` side effect.
"""
import logging
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dr, ops
from .Utils import (bilateral_filter_depth, cluster_poses, compute_mesh_diameter, erode_depth, euler_matrix,
                    make_mesh_tensors, sample_views_icosphere, set_seed)
from .predict_pose_refine import PoseRefinePredictor
from .predict_score import ScorePredictor


class FoundationPose:
    def __init__(self, model_pts, model_normals, symmetry_tfs=None, mesh=None, scorer: ScorePredictor = None,
                 refiner: PoseRefinePredictor = None, glctx=None, debug=0, debug_dir="/tmp/foundationpose_amd_debug",
                 device="cuda", track_graph=False):
        """track_graph=True replays track_one as one captured hipGraph per frame (same arithmetic; off by default so
        that the call sequence of the reference is followed literally)"""
        self.gt_pose = None
        self.track_graph = bool(track_graph)
        self._tracker = None
        self._tracker_key = None
        self.ignore_normal_flip = True
        self.debug = debug
        self.debug_dir = debug_dir
        if debug >= 1:
            os.makedirs(debug_dir, exist_ok=True)
        self.device = torch.device(device)
        self.reset_object(model_pts, model_normals, symmetry_tfs=symmetry_tfs, mesh=mesh)
        self.make_rotation_grid(min_n_views=40, inplane_step=60)
        self.glctx = glctx
        self.scorer = scorer if scorer is not None else ScorePredictor(device=device)
        self.refiner = refiner if refiner is not None else PoseRefinePredictor(device=device)
        self.pose_last = None  # used for tracking; w.r.t. the centred mesh

    # ------------------------------------------------------------------ estimater.py:44-78
    def reset_object(self, model_pts, model_normals, symmetry_tfs=None, mesh=None):
        self._tracker = None   # a captured tracking graph holds the previous object's mesh
        max_xyz = np.asarray(mesh.vertices).max(axis=0)
        min_xyz = np.asarray(mesh.vertices).min(axis=0)
        self.model_center = (min_xyz + max_xyz) / 2
        self.mesh_ori = mesh.copy()
        mesh = mesh.copy()
        mesh.vertices = np.asarray(mesh.vertices) - self.model_center.reshape(1, 3)
        model_pts = np.asarray(mesh.vertices)
        self.diameter = compute_mesh_diameter(model_pts=model_pts, n_sample=10000)
        self.vox_size = max(self.diameter / 20.0, 0.003)
        logging.info(f"self.diameter:{self.diameter}, vox_size:{self.vox_size}")
        self.dist_bin = self.vox_size / 2
        self.angle_bin = 20
        self.max_xyz = model_pts.max(axis=0)
        self.min_xyz = model_pts.min(axis=0)
        self.pts = torch.tensor(model_pts, dtype=torch.float32, device=self.device)
        nrm = np.asarray(model_normals if model_normals is not None else mesh.vertex_normals)
        self.normals = F.normalize(torch.tensor(nrm, dtype=torch.float32, device=self.device), dim=-1)
        self.mesh_path = None
        self.mesh = mesh
        self.mesh_tensors = make_mesh_tensors(self.mesh, device=self.device)
        if symmetry_tfs is None:
            self.symmetry_tfs = torch.eye(4, device=self.device).float()[None]
        else:
            self.symmetry_tfs = torch.as_tensor(symmetry_tfs, device=self.device, dtype=torch.float)
        logging.info("reset done")

    def get_tf_to_centered_mesh(self):
        tf_to_center = torch.eye(4, dtype=torch.float, device=self.device)
        tf_to_center[:3, 3] = -torch.as_tensor(self.model_center, device=self.device, dtype=torch.float)
        return tf_to_center

    # ------------------------------------------------------------------ estimater.py:88-102
    def to_device(self, s="cuda:0"):
        self.device = torch.device(s)
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v) or isinstance(v, nn.Module):
                self.__dict__[k] = v.to(s)
        for k in list(self.mesh_tensors):
            if torch.is_tensor(self.mesh_tensors[k]):
                self.mesh_tensors[k] = self.mesh_tensors[k].to(s)
        self.mesh_tensors.pop("_handle", None)  # rebuilt lazily against the moved tensors
        if self.refiner is not None:
            self.refiner.model.to(s)
        if self.scorer is not None:
            self.scorer.model.to(s)
        if self.glctx is not None:
            self.glctx = dr.RasterizeCudaContext(s)

    # ------------------------------------------------------------------ estimater.py:106-124
    def make_rotation_grid(self, min_n_views=40, inplane_step=60):
        cam_in_obs = sample_views_icosphere(n_views=min_n_views)
        rot_grid = []
        for i in range(len(cam_in_obs)):
            for inplane_rot in np.deg2rad(np.arange(0, 360, inplane_step)):
                cam_in_ob = cam_in_obs[i] @ euler_matrix(0, 0, inplane_rot)
                rot_grid.append(np.linalg.inv(cam_in_ob))
        rot_grid = np.asarray(rot_grid)
        rot_grid = cluster_poses(30, 99999, rot_grid, self.symmetry_tfs.data.cpu().numpy())
        rot_grid = np.asarray(rot_grid)
        logging.info(f"after cluster, rot_grid:{rot_grid.shape}")
        self.rot_grid = torch.as_tensor(rot_grid, device=self.device, dtype=torch.float)

    def generate_random_pose_hypo(self, K, rgb, depth, mask, scene_pts=None):
        ob_in_cams = self.rot_grid.clone()
        center = self.guess_translation(depth=depth, mask=mask, K=K)
        ob_in_cams[:, :3, 3] = torch.as_tensor(center, device=self.device, dtype=torch.float).reshape(1, 3)
        return ob_in_cams

    # ------------------------------------------------------------------ estimater.py:137-156
    def guess_translation(self, depth, mask, K):
        """Initial translation of every hypothesis (semantics of estimater.py:137-156): the ray through the centre of the
        mask's bounding box, at the median of the valid depths inside the mask; zeros when the mask or the valid set is
        empty.  Computed on the device: the depth map never leaves HBM (the reference does this in numpy)."""
        d = torch.as_tensor(depth, device=self.device, dtype=torch.float)
        m = torch.as_tensor(np.asarray(mask) if not torch.is_tensor(mask) else mask, device=self.device) > 0
        rows, cols = torch.nonzero(m.any(dim=1)).reshape(-1), torch.nonzero(m.any(dim=0)).reshape(-1)
        if rows.numel() == 0:
            logging.info("mask is all zero")
            return np.zeros((3))
        z = d[m & (d >= 0.001)]
        if z.numel() == 0:
            logging.info("valid is empty")
            return np.zeros((3))
        zs = torch.sort(z).values
        n = zs.numel()
        stats = torch.stack([rows[0], rows[-1], cols[0], cols[-1]]).to(torch.float64)
        mid = torch.stack([zs[(n - 1) // 2], zs[n // 2]])          # numpy's median: mean of the two middle values
        v0, v1, u0, u1 = stats.tolist()                            # six scalars cross PCIe, not a 640x480 image
        lo, hi = mid.tolist()
        zc = float(np.float32(np.float32(lo) + np.float32(hi)) / np.float32(2.0)) if lo != hi else lo
        center = (np.linalg.inv(K) @ np.asarray([(u0 + u1) / 2.0, (v0 + v1) / 2.0, 1]).reshape(3, 1)) * zc   # estimater.py:149
        return center.reshape(3)

    # ------------------------------------------------------------------ estimater.py:159-240
    def register(self, K, rgb, depth, ob_mask, ob_id=None, glctx=None, iteration=5):
        set_seed(0)
        if self.glctx is None:
            self.glctx = glctx if glctx is not None else dr.RasterizeCudaContext(self.device)
        depth_t = torch.as_tensor(depth, device=self.device, dtype=torch.float).contiguous()
        depth_t = ops.erode_depth(depth_t, radius=2)
        depth_t = ops.bilateral_filter_depth(depth_t, radius=2)
        ob_mask = np.asarray(ob_mask.data.cpu().numpy() if torch.is_tensor(ob_mask) else ob_mask)
        mask_t = torch.as_tensor(ob_mask, device=self.device) > 0
        if int(((depth_t >= 0.001) & mask_t).sum()) < 4:
            logging.info("valid too small, return")
            pose = np.eye(4)
            pose[:3, 3] = self.guess_translation(depth=depth_t, mask=mask_t, K=K)
            return pose
        self.H, self.W = int(depth_t.shape[0]), int(depth_t.shape[1])
        self.K = K
        self.ob_id = ob_id
        self.ob_mask = ob_mask
        poses = self.generate_random_pose_hypo(K=K, rgb=rgb, depth=depth_t, mask=mask_t, scene_pts=None)
        xyz_map = ops.depth_to_xyz(depth_t, K, zfar=float("inf"), f64_internal=True)  # depth2xyzmap (numpy variant)
        poses, vis = self.refiner.predict(mesh=self.mesh, mesh_tensors=self.mesh_tensors, rgb=rgb, depth=depth_t, K=K,
                                          ob_in_cams=poses, normal_map=None, xyz_map=xyz_map, glctx=self.glctx,
                                          mesh_diameter=self.diameter, iteration=iteration, get_vis=self.debug >= 2,
                                          shared_translation=True)   # generate_random_pose_hypo: one centre for the whole grid
        scores, vis = self.scorer.predict(mesh=self.mesh, rgb=rgb, depth=depth_t, K=K, ob_in_cams=poses,
                                          normal_map=None, mesh_tensors=self.mesh_tensors, glctx=self.glctx,
                                          mesh_diameter=self.diameter, get_vis=self.debug >= 2)
        ids = torch.as_tensor(scores).argsort(descending=True)
        scores = scores[ids]
        poses = poses[ids]
        best_pose = poses[0] @ self.get_tf_to_centered_mesh()
        self.pose_last = poses[0]
        self.best_id = ids[0]
        self.poses = poses
        self.scores = scores
        return best_pose.data.cpu().numpy()

    def compute_add_err_to_gt_pose(self, poses):
        """stub in the reference as well (estimater.py:243-247)"""
        return -torch.ones(len(poses), device=self.device, dtype=torch.float)

    # ------------------------------------------------------------------ estimater.py:250-268
    def track_one(self, rgb, depth, K, iteration, extra={}):
        if self.pose_last is None:
            logging.info("Please init pose by register first")
            raise RuntimeError
        if self.track_graph:
            # same arithmetic, one hipGraph launch per frame (foundationpose_amd/graphs.py); re-captured when the
            # frame size, intrinsics or iteration count change
            key = (tuple(np.asarray(depth).shape[:2]), np.asarray(K, dtype=np.float64).tobytes(), int(iteration))
            if self._tracker is None or self._tracker_key != key:
                from .graphs import GraphedTracker
                H, W = key[0]
                self._tracker = GraphedTracker(self.refiner, self.mesh_tensors, self.diameter, K, H, W, n_hyp=1,
                                               iteration=iteration, device=self.device).capture()
                self._tracker_key = key
            pose = self._tracker.step(rgb, depth, self.pose_last.reshape(1, 4, 4)).clone()
            self.pose_last = pose
            return (pose @ self.get_tf_to_centered_mesh()).data.cpu().numpy().reshape(4, 4)
        depth_t = torch.as_tensor(depth, device=self.device, dtype=torch.float).contiguous()
        depth_t = ops.erode_depth(depth_t, radius=2)
        depth_t = ops.bilateral_filter_depth(depth_t, radius=2)
        xyz_map = ops.depth_to_xyz(depth_t, K, zfar=float("inf"), f64_internal=False)  # depth2xyzmap_batch variant
        pose, vis = self.refiner.predict(mesh=self.mesh, mesh_tensors=self.mesh_tensors, rgb=rgb, depth=depth_t, K=K,
                                         ob_in_cams=self.pose_last.reshape(1, 4, 4), normal_map=None, xyz_map=xyz_map,
                                         mesh_diameter=self.diameter, glctx=self.glctx, iteration=iteration,
                                         get_vis=self.debug >= 2)
        if self.debug >= 2:
            extra["vis"] = vis
        self.pose_last = pose
        return (pose @ self.get_tf_to_centered_mesh()).data.cpu().numpy().reshape(4, 4)

    track = track_one  # the north-star calls it track(); the reference method is track_one
