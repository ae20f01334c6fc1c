"""ScorePredictor -- drop-in for learning/training/predict_score.py:117-226."""

import numpy as np
import torch

from . import ops
from .Utils import get_mesh_handle, make_mesh_tensors
from .engine import ScorePlan
from .h5_dataset import ScoreMultiPairH5Dataset
from .pose_dataset import BatchPoseData
from .graphs import GraphCache, PartGraphs
from .overlap import SubBatches
from .predict_pose_refine import _Cfg, load_run
from .score_network import ScoreNetMultiPair

_SCORE_DEFAULTS = dict(use_normal=False, use_BN=False, zfar=np.inf, c_in=4, normalize_xyz=False, crop_ratio=1.2)


def make_crop_data_batch(render_size, ob_in_cams, mesh, rgb, depth, K, crop_ratio, normal_map=None, mesh_diameter=None,
                         glctx=None, mesh_tensors=None, dataset=None, cfg=None, AB=None, workspace=None):
    """Reference: predict_score.py:56-114 + TripletH5Dataset.transform_depth_to_xyzmap (h5_dataset.py:137-170).
    The observed xyz is rebuilt from the depth crop through the frame (3 nearest-neighbour hops) inside one kernel
    instead of two (N,480,640[,3]) intermediates (~1.2 GB at N=252).  AB: optional (2N,6,h,w) destination; workspace: optional
    caller-owned rasteriser scratch (ops.workspace_bytes) -- a captured hipGraph owns its scratch."""
    H, W = depth.shape[:2]
    handle = get_mesh_handle(mesh_tensors)
    poseA = torch.as_tensor(ob_in_cams, dtype=torch.float, device=handle.device).reshape(-1, 4, 4).contiguous()
    N = poseA.shape[0]
    oh, ow = int(cfg["input_resize"][0]), int(cfg["input_resize"][1])
    tf_to_crops, bbox2d = ops.crop_windows(poseA, K, mesh_diameter, crop_ratio, (render_size[1], render_size[0]))
    if AB is None:
        AB = torch.empty((2 * N, 6, oh, ow), dtype=torch.float32, device=handle.device)
    normalize = bool(cfg["normalize_xyz"])
    for b in range(0, N, 4096):
        e = min(N, b + 4096)
        ops.render_crops(handle, poseA[b:e], bbox2d[b:e], K, H, W, out_hw=(oh, ow), mesh_diameter=mesh_diameter,
                         xyz_thr=0.1, normalize_xyz=normalize, A_out=AB[b:e], workspace=workspace)
        ops.warp_crops(rgb, None, depth, tf_to_crops[b:e], K, poseA[b:e], mesh_diameter, ops.MODE_SCORE,
                       normalize_xyz=normalize, out_hw=(oh, ow), B_out=AB[N + b:N + e])
    Ks = torch.as_tensor(np.asarray(K, dtype=np.float64), dtype=torch.float, device=handle.device).reshape(1, 3, 3).expand(N, 3, 3)
    mesh_diameters = torch.ones((N,), dtype=torch.float, device=handle.device) * float(mesh_diameter)
    batch = BatchPoseData(rgbAs=AB[:N, :3], rgbBs=AB[N:, :3], xyz_mapAs=AB[:N, 3:], xyz_mapBs=AB[N:, 3:], poseA=poseA,
                          tf_to_crops=tf_to_crops, Ks=Ks, mesh_diameters=mesh_diameters)
    batch.AB = AB
    if dataset is not None:
        batch = dataset.transform_batch(batch, H_ori=H, W_ori=W, bound=1)
    return batch


class ScorePredictor:
    run_name = "2024-01-11-20-02-45"

    def __init__(self, amp=True, cfg=None, state_dict=None, weights_root=None, device="cuda", precision=None,
                 channels_last=True, n_streams=2, graph="auto"):
        self.sub = SubBatches(n_streams)      # hypothesis sub-batches that run concurrently (overlap.py); 1 = none
        self.graph = graph                    # hipGraph replay of predict(): True / "auto" / False as PoseRefinePredictor
        self._graphs = GraphCache()
        if precision is None:
            precision = "fp16" if amp else "fp32"
        self.amp = precision != "fp32"
        if cfg is None or state_dict is None:
            cfg, state_dict, ckpt_dir = load_run(self.run_name, weights_root)
        else:
            ckpt_dir = None
        self.cfg = _Cfg(cfg)
        self.cfg["ckpt_dir"] = ckpt_dir
        self.cfg["enable_amp"] = True
        for k, v in _SCORE_DEFAULTS.items():  # predict_score.py:131-142
            if k not in self.cfg or (k == "crop_ratio" and self.cfg[k] is None):
                self.cfg[k] = v
        # use_normal=True only makes the reference render normals it then drops (predict_score.py:78,107-108: normalAs =
        # normalBs = None); accepted and without effect here.  c_in has to be 6 for the same reason as in the refiner.
        if self.cfg["c_in"] != 6:
            raise NotImplementedError("c_in must be 6: the scorer is fed cat([rgb, xyz]) (predict_score.py:188-189)")
        self.dataset = ScoreMultiPairH5Dataset(cfg=self.cfg, mode="test", h5_file=None, max_num_key=1)
        self.device = torch.device(device)
        self.precision = precision
        self._plan_opts = dict(precision=precision, channels_last=channels_last)
        self.model = ScoreNetMultiPair(cfg=self.cfg, c_in=self.cfg["c_in"])
        self.model.load_state_dict(state_dict)
        self.model.to(self.device).eval()
        self._plan = None
        self.small_calls = True      # engine.ScorePlan.small_calls: False = a few-hypothesis call runs the kernels of a large one (dist.py)

    def plan(self):
        dev = next(self.model.parameters()).device
        if self._plan is None or self._plan_dev != dev:
            self._plan = ScorePlan(self.model, dev, **self._plan_opts)
            self._plan_dev = dev
        self._plan.small_calls = bool(self.small_calls)
        return self._plan

    @torch.inference_mode()
    def _graphed_features(self, rgb_t, depth_t, K, poses, mesh, mesh_tensors, mesh_diameter, glctx, mode):
        """the per-hypothesis half of predict() (crops + encoder + self-attention + pooling) as ONE linear hipGraph per
        sub-batch, replayed on the sub-batch streams (graphs.PartGraphs): -> the static (N, 512) feature buffer, or None when
        this call is to run eagerly.  Bit-identical to the eager launches (tests/test_gpu_parity.py)."""
        plan = self.plan()
        if not plan.hip or mode is False or mode is None or torch.cuda.is_current_stream_capturing() or ops.KernelTimers.active is not None:
            return None
        dev, N = poses.device, int(poses.shape[0])
        handle = get_mesh_handle(mesh_tensors)
        parts = tuple(self.sub.parts(N, dev))
        H, W = int(depth_t.shape[0]), int(depth_t.shape[1])
        key = (N, H, W, np.asarray(K, dtype=np.float64).tobytes(), id(handle), float(mesh_diameter), parts, bool(self.sub.serial), dev.index,
               bool(plan.small_calls))

        def build():
            oh, ow = int(self.cfg["input_resize"][0]), int(self.cfg["input_resize"][1])
            g = dict(rgb=torch.empty_like(rgb_t), depth=torch.empty_like(depth_t), poses=torch.empty_like(poses), mesh=mesh_tensors,
                     feats=torch.empty((N, 512), dtype=plan.dtype, device=dev),
                     ws=[torch.empty(max(16, ops.workspace_bytes(b - a, handle.V, handle.T, oh, ow)), dtype=torch.uint8, device=dev)
                         for a, b in parts])
            g["rgb"].copy_(rgb_t); g["depth"].copy_(depth_t); g["poses"].copy_(poses)

            normalize = bool(self.cfg["normalize_xyz"])

            def body(h):
                # the launches of make_crop_data_batch without its BatchPoseData bookkeeping (whose intrinsics / diameter tensors
                # are host -> device copies, which a stream capture does not allow): crop windows, rendered crop, observed crop
                a, b = parts[h]
                n = b - a
                P = g["poses"][a:b]
                AB = torch.empty((2 * n, 6, oh, ow), dtype=plan.dtype, device=dev)
                tf_to_crops, bbox2d = ops.crop_windows(P, K, mesh_diameter, self.cfg["crop_ratio"], (ow, oh))
                ops.render_crops(handle, P, bbox2d, K, H, W, out_hw=(oh, ow), mesh_diameter=mesh_diameter, xyz_thr=0.1,
                                 normalize_xyz=normalize, A_out=AB[:n], workspace=g["ws"][h])
                ops.warp_crops(g["rgb"], None, g["depth"], tf_to_crops, K, P, mesh_diameter, ops.MODE_SCORE, normalize_xyz=normalize,
                               out_hw=(oh, ow), B_out=AB[n:])
                plan.features(AB, slot=h, out=g["feats"][a:b])
            g["graphs"] = PartGraphs(self.sub, dev, len(parts), body)
            return g
        g = self._graphs.get(key, mode, build)
        if g is None:
            return None
        g["rgb"].copy_(rgb_t); g["depth"].copy_(depth_t); g["poses"].copy_(poses)
        g["graphs"].replay()
        return g["feats"]

    @torch.inference_mode()
    def predict(self, rgb, depth, K, ob_in_cams, normal_map=None, get_vis=False, mesh=None, mesh_tensors=None,
                glctx=None, mesh_diameter=None, feature_exchange=None, graph=None):
        """-> (scores (N,) f32 device tensor = logit + 100, vis).  ``feature_exchange``: optional callable applied to the
        pooled per-hypothesis features before the cross-hypothesis attention (multi-GPU all-gather hook, dist.py).  graph: None =
        the predictor's setting; True / "auto" / False: hipGraph replay of the per-hypothesis half (_graphed_features)."""
        plan = self.plan()
        dev = self._plan_dev
        if mesh_tensors is None:
            mesh_tensors = make_mesh_tensors(mesh, device=dev)
        poses = torch.as_tensor(ob_in_cams, dtype=torch.float, device=dev).reshape(-1, 4, 4).contiguous()
        N = poses.shape[0]
        rgb_t = torch.as_tensor(rgb, device=dev).to(torch.float).contiguous()
        depth_t = torch.as_tensor(depth, device=dev, dtype=torch.float).contiguous()
        oh, ow = int(self.cfg["input_resize"][0]), int(self.cfg["input_resize"][1])
        # the encoder + per-hypothesis attention see one hypothesis at a time: sub-batches on concurrent streams
        # (overlap.py), joined before the cross-hypothesis attention, which needs all N feature rows
        feats = None if get_vis else self._graphed_features(rgb_t, depth_t, K, poses, mesh, mesh_tensors, mesh_diameter, glctx,
                                                            self.graph if graph is None else graph)
        batches = []
        if feats is None:
            parts = self.sub.parts(N, dev)
            feats = torch.empty((N, 512), dtype=plan.dtype, device=dev)
            streams = self.sub.streams(dev, len(parts))
            self.sub.fork(streams)
            for h, (a, b) in enumerate(parts):
                with torch.cuda.stream(streams[h]):
                    AB = torch.empty((2 * (b - a), 6, oh, ow), dtype=plan.dtype, device=dev)
                    batch = make_crop_data_batch(self.cfg["input_resize"], poses[a:b], mesh, rgb_t, depth_t, K,
                                                 crop_ratio=self.cfg["crop_ratio"], glctx=glctx, mesh_tensors=mesh_tensors,
                                                 dataset=self.dataset, cfg=self.cfg, mesh_diameter=mesh_diameter, AB=AB)
                    plan.features(batch.AB, slot=h, out=feats[a:b])
                    batches.append(batch)
            self.sub.join(streams)
        if feature_exchange is not None:
            feats = feature_exchange(feats)
        # bs == N in the reference (predict_score.py:186), so its pairwise tournament always ends after one round
        logits = plan.head(feats, L=feats.shape[0]).reshape(-1)
        scores = logits + 100  # predict_score.py:209
        if get_vis:
            # debug canvas (predict_score.py:27-52, :219-223): the crops of all hypotheses, best score first
            from .vis import crop_rows_canvas
            ids = scores.argsort(descending=True).cpu().numpy()
            A = np.concatenate([bt.AB[: bt.AB.shape[0] // 2].float().cpu().numpy() for bt in batches], 0)
            B = np.concatenate([bt.AB[bt.AB.shape[0] // 2:].float().cpu().numpy() for bt in batches], 0)
            return scores, crop_rows_canvas(A, B, ids=ids)
        return scores, None
