"""PoseRefinePredictor -- drop-in for learning/training/predict_pose_refine.py:93-295.

``predict`` keeps the reference signature and return types.  Per iteration the reference launches ~60 kernels
(crop tf, nvdiffrast x7, kornia warps, dataset transform, concat, pose update); here it is four launches of
libfp_amd.so (fp_crop_windows, fp_render_crops, fp_warp_crops, fp_pose_update) around the network plan, with
no host round trip inside the loop.
"""
import os

import numpy as np
import torch

from . import ops
from .Utils import get_mesh_handle, make_mesh_tensors
from .engine import RefinePlan
from .h5_dataset import PoseRefinePairH5Dataset
from .graphs import GraphCache, PartGraphs
from .overlap import SubBatches
from .pose_dataset import BatchPoseData
from .refine_network import RefineNet

_REFINE_DEFAULTS = dict(use_normal=False, use_mask=False, use_BN=False, c_in=4, crop_ratio=1.2, n_view=1,
                        trans_rep="tracknet", rot_rep="axis_angle", zfar=3, normalize_xyz=False, normal_uint8=False)


class _Cfg(dict):
    """dict with attribute access (stands in for OmegaConf's DictConfig, which is not available here)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def load_run(run_name, weights_root=None):
    """weights/<run_name>/{config.yml, model_best.pth} as in predict_pose_refine.py:97-141."""
    import yaml
    root = weights_root or os.environ.get("FOUNDATIONPOSE_WEIGHTS") or os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "weights")
    cfg_path = os.path.join(root, run_name, "config.yml")
    ckpt_path = os.path.join(root, run_name, "model_best.pth")
    if not (os.path.exists(cfg_path) and os.path.exists(ckpt_path)):
        raise FileNotFoundError(
            f"pretrained run '{run_name}' not found under {root}; pass cfg= and state_dict= explicitly "
            f"(e.g. foundationpose_amd.weights.random_state_dict) or set FOUNDATIONPOSE_WEIGHTS")
    with open(cfg_path) as f:
        cfg = yaml.safe_load(f)
    ckpt = torch.load(ckpt_path, map_location="cpu")
    if "model" in ckpt:
        ckpt = ckpt["model"]
    return cfg, ckpt, ckpt_path


def make_crop_data_batch(render_size, ob_in_cams, mesh, rgb, depth, K, crop_ratio, xyz_map, normal_map=None,
                         mesh_diameter=None, cfg=None, glctx=None, mesh_tensors=None, dataset=None, AB=None, depth_hw=None):
    """Reference: predict_pose_refine.py:26-89.  Returns a BatchPoseData whose rgbAs/xyz_mapAs/rgbBs/xyz_mapBs are
    views into one (2N,6,h,w) network-input buffer (``batch.AB``): A = rendered hypothesis, B = observed crop."""
    H, W = depth_hw if depth is None else depth.shape[:2]
    handle = get_mesh_handle(mesh_tensors)
    poseA = torch.as_tensor(ob_in_cams, dtype=torch.float, device=handle.device).reshape(-1, 4, 4).contiguous()
    N = poseA.shape[0]
    oh, ow = int(cfg["input_resize"][0]), int(cfg["input_resize"][1])
    tf_to_crops, bbox2d = ops.crop_windows(poseA, K, mesh_diameter, crop_ratio, (render_size[1], render_size[0]))
    if N == 2:
        # reference broadcasting quirk: with exactly two poses transform_pts pairs pose i with
        # corner i, so both hypotheses are rendered with [umin_0, vmin_0, umax_1, vmax_1]
        bbox2d = torch.stack([bbox2d[0, 0], bbox2d[0, 1], bbox2d[1, 2], bbox2d[1, 3]])[None].expand(2, 4).contiguous()
    if AB is None:
        AB = torch.empty((2 * N, 6, oh, ow), dtype=torch.float32, device=handle.device)
    normalize = bool(cfg["normalize_xyz"])
    for b in range(0, N, 4096):
        e = min(N, b + 4096)
        ops.render_crops(handle, poseA[b:e], bbox2d[b:e], K, H, W, out_hw=(oh, ow), mesh_diameter=mesh_diameter,
                         xyz_thr=0.001, normalize_xyz=normalize, A_out=AB[b:e])
        ops.warp_crops(rgb, xyz_map, None, tf_to_crops[b:e], K, poseA[b:e], mesh_diameter, ops.MODE_REFINE,
                       normalize_xyz=normalize, out_hw=(oh, ow), B_out=AB[N + b:N + e])
    normalAs = normalBs = None
    if cfg.get("use_normal", False):
        # predict_pose_refine.py:50,58,75-76: the rendered normals and the frame's normal map go through the SAME nearest
        # warp by tf_to_crops (the rendered crop is treated as if it were a frame -- reference behaviour, kept) into
        # BatchPoseData.normalAs / normalBs.  The reference never concatenates them into the network input (:187-188 take rgb
        # and xyz only), so this is bookkeeping off the hot path: plain torch ops, and refine_part() skips it.
        if normal_map is None:
            raise ValueError("use_normal=True needs a normal_map (the reference fails in torch.as_tensor(None) here)")
        from .Utils import warp_perspective_nearest
        nr = torch.cat([ops.render_crops(handle, poseA[b:b + 4096], bbox2d[b:b + 4096], K, H, W, out_hw=(oh, ow), mesh_diameter=mesh_diameter,
                                         xyz_thr=0.001, normalize_xyz=normalize, want=("normal",))["normal"] for b in range(0, N, 4096)])
        normalAs = warp_perspective_nearest(nr.permute(0, 3, 1, 2).contiguous(), tf_to_crops, render_size)
        nm = torch.as_tensor(normal_map, dtype=torch.float, device=handle.device).permute(2, 0, 1)[None].expand(N, -1, -1, -1)
        normalBs = warp_perspective_nearest(nm, tf_to_crops, render_size)
    Ks = torch.as_tensor(np.asarray(K, dtype=np.float64), device=handle.device, dtype=torch.float).reshape(1, 3, 3)
    mesh_diameters = torch.ones((N,), dtype=torch.float, device=handle.device) * float(mesh_diameter)
    batch = BatchPoseData(rgbAs=AB[:N, :3], rgbBs=AB[N:, :3], xyz_mapAs=AB[:N, 3:], xyz_mapBs=AB[N:, 3:], poseA=poseA,
                          normalAs=normalAs, normalBs=normalBs, tf_to_crops=tf_to_crops, Ks=Ks, mesh_diameters=mesh_diameters)
    batch.AB = AB
    batch.bbox2d = bbox2d
    if dataset is not None:
        batch = dataset.transform_batch(batch=batch, H_ori=H, W_ori=W, bound=1)
    return batch


def resolve_shared_translation(ob_in_cams, flag):
    """-> bool: may the first refine iteration treat the hypotheses as sharing one translation (PoseRefinePredictor.predict)?
    flag True: the caller says so -- and is checked: host poses on the host, a device tensor with one reduction on the device
    (one scalar read back; skipped only inside a stream capture, where nothing may synchronise); None: decided from host poses
    -- compared as the float32 values that are uploaded -- and False for a device tensor, which is not read back for a mere
    optimisation; False: no."""
    on_host = isinstance(ob_in_cams, np.ndarray) or (torch.is_tensor(ob_in_cams) and ob_in_cams.device.type == "cpu") or \
        isinstance(ob_in_cams, (list, tuple))
    if not on_host:
        if flag and torch.is_tensor(ob_in_cams) and ob_in_cams.is_cuda and not torch.cuda.is_current_stream_capturing():
            t = ob_in_cams.reshape(-1, 4, 4)[:, :3, 3].float()
            if t.shape[0] > 1 and not bool((t == t[:1]).all().item()):
                raise ValueError("shared_translation=True, but the hypotheses do not have one translation")
        return bool(flag)
    t = np.asarray(ob_in_cams, dtype=np.float32).reshape(-1, 4, 4)[:, :3, 3]
    same = bool(t.shape[0] > 1 and (t == t[:1]).all())
    if flag and t.shape[0] > 1 and not same:
        raise ValueError("shared_translation=True, but the hypotheses do not have one translation")
    return same if flag is None else bool(flag) and same


class PoseRefinePredictor:
    run_name = "2023-10-28-18-33-37"

    def __init__(self, cfg=None, state_dict=None, weights_root=None, device="cuda", precision="fp16", channels_last=True,
                 n_streams=2, graph="auto"):
        """precision='fp16': the reference's deployed autocast configuration on libfp_amd.so (engine.py);
        'fp32': fp32 torch ops, no autocast (`amp=False` in the reference).  n_streams: hypothesis sub-batches that run
        concurrently (overlap.py); 1 = one launch sequence over the whole batch.  graph: hipGraph replay of predict()'s refine
        loop (fp16 plan only): True = capture at the first call of a (shape, intrinsics, mesh, iteration) key, "auto" = at the
        second (graphs.GraphCache), False = always eager launches; per call: predict(graph=...)"""
        self.sub = SubBatches(n_streams)
        self.graph = graph
        self._graphs = GraphCache()
        self.amp = precision != "fp32"
        if cfg is None or state_dict is None:
            cfg, state_dict, ckpt_dir = load_run(self.run_name, weights_root)
        else:
            ckpt_dir = None
        self.cfg = _Cfg(cfg)
        self.cfg["ckpt_dir"] = ckpt_dir
        self.cfg["enable_amp"] = True
        for k, v in _REFINE_DEFAULTS.items():  # backward-compat defaults, predict_pose_refine.py:107-130
            if k not in self.cfg or (k == "crop_ratio" and self.cfg[k] is None):
                self.cfg[k] = v
        if isinstance(self.cfg["zfar"], str) and "inf" in self.cfg["zfar"].lower():
            self.cfg["zfar"] = np.inf
        for k in ("input_resize", "trans_normalizer", "rot_normalizer"):
            if k not in self.cfg:
                raise KeyError(f"refiner cfg lacks required key '{k}'")
        # use_normal=True is accepted: in the reference it only adds normalAs / normalBs to the BatchPoseData of
        # make_crop_data_batch (above); predict() builds A and B from rgb and xyz alone (predict_pose_refine.py:187-188),
        # so the networks -- and refine_part() here -- never see the normals.  For the same reason c_in has to be 6: with any
        # other value the reference's own forward fails on the 6-channel A / B (channel mismatch in the first conv).
        if self.cfg["c_in"] != 6:
            raise NotImplementedError("c_in must be 6: predict() feeds cat([rgb, xyz]) to the network whatever use_normal says "
                                      "(predict_pose_refine.py:187-188); the reference fails in its first conv otherwise")
        self.dataset = PoseRefinePairH5Dataset(cfg=self.cfg, h5_file="", mode="test")
        self.device = torch.device(device)
        self.precision = precision
        self._plan_opts = dict(precision=precision, channels_last=channels_last)
        self.model = RefineNet(cfg=self.cfg, c_in=self.cfg["c_in"])
        self.model.load_state_dict(state_dict)
        self.model.to(self.device).eval()
        self._plan = None
        self.small_calls = True      # engine.RefinePlan.small_calls: False = a few-hypothesis call runs the kernels of a large one (dist.py)
        self.last_trans_update = None
        self.last_rot_update = None

    def plan(self):
        dev = next(self.model.parameters()).device
        if self._plan is None or self._plan_dev != dev:
            self._plan = RefinePlan(self.model, dev, **self._plan_opts)
            self._plan.two_stream_heads = self.sub.n_streams > 1      # n_streams=1 switches every use of the side stream off
            self._plan_dev = dev
            if self._plan.hip and self._plan.two_stream_heads and dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                # create and probe the side stream NOW (spin kernels + the rasteriser canary synchronise): the first small call -- the
                # reference's track_one -- must not pay for it, and a capture must never meet an unprobed stream (engine._head_side_stream)
                from .overlap import reserve_streams
                reserve_streams(dev, 1)
        self._plan.small_calls = bool(self.small_calls)
        return self._plan

    def _loop_constants(self):
        oh, ow = int(self.cfg["input_resize"][0]), int(self.cfg["input_resize"][1])
        tn = self.cfg["trans_normalizer"]
        tn = [float(tn)] * 3 if isinstance(tn, (int, float)) else [float(v) for v in tn]
        return oh, ow, tn, bool(self.cfg["normalize_xyz"])

    def refine_part(self, slot, rows, rgb_t, xyz_t, poses, K, H, W, mesh_handle, mesh_diameter, iterations, outs, workspace=None,
                    state=None, shared_translation=False):
        """Iterations `iterations` (a range) of the refine loop for the hypotheses rows=(a, b) of `poses`, on the CURRENT
        stream, with the activation-buffer set `slot`: per iteration fp_crop_windows -> fp_render_crops (A) +
        fp_warp_crops (B) -> RefineNet plan -> fp_pose_update.  outs = (poses_out (N,4,4), trans_delta (N,3), rot_delta
        (N,3,3), n_iterations_total): the last iteration writes rows a..b of them.  state: what the previous call for this
        part returned (None for the first iteration).  shared_translation: the hypotheses a..b of `poses` all have the same
        translation (register(): estimater.py:132-133 puts every rotation of the grid at the guessed centre), so in
        iteration 0 they share one crop window and one observed crop: it is warped once and the stem of the fp16 plan
        encodes it once (engine._HipEncoder, bit-identical to 252 copies).  -> state"""
        plan = self.plan()
        a, b = rows
        n = b - a
        oh, ow, tn, normalize = self._loop_constants()
        poses_out, trans_delta, rot_delta, total = outs
        if state is None:
            state = dict(P=poses[a:b], AB=torch.empty((2 * n, 6, oh, ow), dtype=plan.dtype, device=poses.device), raw=None)
        for it in iterations:
            last = it + 1 == total
            P, AB = state["P"], state["AB"]
            tf_to_crops, bbox2d = ops.crop_windows(P, K, mesh_diameter, self.cfg["crop_ratio"], (ow, oh))
            if poses.shape[0] == 2:
                # reference broadcasting quirk: with exactly two poses transform_pts pairs pose i with
                # corner i, so both hypotheses are rendered with [umin_0, vmin_0, umax_1, vmax_1]
                bbox2d = torch.stack([bbox2d[0, 0], bbox2d[0, 1], bbox2d[1, 2], bbox2d[1, 3]])[None].expand(2, 4).contiguous()
            ops.render_crops(mesh_handle, P, bbox2d, K, H, W, out_hw=(oh, ow), mesh_diameter=mesh_diameter, xyz_thr=0.001,
                             normalize_xyz=normalize, A_out=AB[:n], workspace=workspace)
            shared = bool(shared_translation) and it == 0 and n > 1 and plan.hip
            ops.warp_crops(rgb_t, xyz_t, None, tf_to_crops[:1] if shared else tf_to_crops, K, P[:1] if shared else P, mesh_diameter,
                           ops.MODE_REFINE, normalize_xyz=normalize, out_hw=(oh, ow), B_out=AB[n:n + 1] if shared else AB[n:])
            raw = plan(AB[:n + 1], slot=slot, shared_b=True) if shared else plan(AB, slot=slot)
            state["raw"] = raw
            state["P"] = ops.pose_update(raw["trans"], raw["rot"], P, rot_rep=self.cfg["rot_rep"], normalize_xyz=normalize,
                                         trans_normalizer=tn, rot_normalizer=float(self.cfg["rot_normalizer"]),
                                         mesh_diameter=float(mesh_diameter), out=poses_out[a:b] if last else None,
                                         trans_delta_out=trans_delta[a:b] if last else None,
                                         rot_delta_out=rot_delta[a:b] if last else None, trans_rep=str(self.cfg["trans_rep"]), K=K,
                                         tf_to_crops=tf_to_crops, input_w=float(self.cfg["input_resize"][0]))
        return state

    def refine_device(self, rgb_t, xyz_t, poses, K, H, W, mesh_handle, mesh_diameter, iteration, workspace=None,
                      shared_translation=False):
        """The refine loop on device tensors only (predict_pose_refine.py:182-235).  No host round trip, no host-side
        tensor creation.  Hypotheses are independent through all iterations, so the parts of `self.sub.parts(N)` run the
        whole loop as independent launch sequences on concurrent streams (overlap.py), issued iteration by iteration and
        joined once at the end.  workspace: optional rasteriser scratch, one uint8 tensor per part.
        -> (poses (N,4,4), trans_delta (N,3) in metres, rot_mat_delta (N,3,3)) of the last iteration, as the reference
        keeps them in last_trans_update / last_rot_update (predict_pose_refine.py:238-239)"""
        self.plan()
        N = poses.shape[0]
        dev = poses.device
        parts = self.sub.parts(N, dev)
        if workspace is not None and torch.is_tensor(workspace):
            workspace = [workspace]
        if workspace is not None and len(workspace) != len(parts):
            raise ValueError(f"refine_device: {len(parts)} parts need {len(parts)} workspaces, got {len(workspace)}")
        outs = self.alloc_outputs(N, dev) + (int(iteration),)
        if iteration <= 0:          # no update happened: the pose unchanged, a zero translation and an identity rotation delta
            outs[0].copy_(poses)
            outs[1].zero_()
            outs[2].copy_(torch.eye(3, dtype=torch.float32, device=dev).expand(N, 3, 3))
        streams = self.sub.streams(dev, len(parts))
        self.sub.fork(streams)
        state = [None] * len(parts)
        for it in range(iteration):
            for h, rows in enumerate(parts):
                with torch.cuda.stream(streams[h]):
                    state[h] = self.refine_part(h, rows, rgb_t, xyz_t, poses, K, H, W, mesh_handle, mesh_diameter, range(it, it + 1),
                                                outs, None if workspace is None else workspace[h], state[h],
                                                shared_translation=shared_translation)
        self.sub.join(streams)
        self._raw_parts = [None if st is None else st["raw"] for st in state]   # last_raw_output
        return outs[:3]

    def refine_graphed(self, rgb_t, xyz_t, poses, K, H, W, mesh_tensors, mesh_diameter, iteration, shared_translation, mode):
        """refine_device as a replay of hipGraphs: ONE linear graph per hypothesis sub-batch holding its whole `iteration`-deep
        launch sequence (~90 launches per iteration), replayed on the sub-batch streams (graphs.PartGraphs) -- the same kernels
        on the same streams in the same order, so the result is bit-identical to the eager loop
        (tests/test_gpu_parity.py::test_graphed_predict_is_the_eager_predict).  Everything a launch addresses by raw pointer is
        static and owned by the cache entry: copies of the frame (rgb, xyz_map) and of the start poses, the outputs, one
        rasteriser scratch per part; intermediates live in the graphs' private pools, activations in the plan (never dropped).
        -> (poses, trans_delta, rot_delta) as fresh tensors, or None when this call is to run eagerly (`mode`, GraphCache)"""
        plan = self.plan()
        if not plan.hip or iteration <= 0 or mode is False or mode is None or torch.cuda.is_current_stream_capturing() \
                or ops.KernelTimers.active is not None:
            return None
        dev, N = poses.device, int(poses.shape[0])
        handle = get_mesh_handle(mesh_tensors)
        parts = tuple(self.sub.parts(N, dev))
        key = (N, int(iteration), H, W, np.asarray(K, dtype=np.float64).tobytes(), id(handle), float(mesh_diameter),
               bool(shared_translation), parts, bool(self.sub.serial), dev.index, bool(plan.small_calls))

        def build():
            oh, ow, _, _ = self._loop_constants()
            g = dict(rgb=torch.empty_like(rgb_t), xyz=torch.empty_like(xyz_t), poses=torch.empty_like(poses), mesh=mesh_tensors,
                     outs=self.alloc_outputs(N, dev) + (int(iteration),), state=[None] * len(parts),
                     ws=[torch.empty(max(16, ops.workspace_bytes(b - a, handle.V, handle.T, oh, ow)), dtype=torch.uint8, device=dev)
                         for a, b in parts])
            g["rgb"].copy_(rgb_t); g["xyz"].copy_(xyz_t); g["poses"].copy_(poses)

            def body(h):
                g["state"][h] = self.refine_part(h, parts[h], g["rgb"], g["xyz"], g["poses"], K, H, W, handle, mesh_diameter,
                                                 range(int(iteration)), g["outs"], g["ws"][h], None, shared_translation=shared_translation)
            g["graphs"] = PartGraphs(self.sub, dev, len(parts), body)
            return g
        g = self._graphs.get(key, mode, build)
        if g is None:
            return None
        g["rgb"].copy_(rgb_t); g["xyz"].copy_(xyz_t); g["poses"].copy_(poses)
        g["graphs"].replay()
        self._raw_parts = [st["raw"] for st in g["state"]]      # tensors of the graphs' pools, refreshed by every replay
        return tuple(t.clone() for t in g["outs"][:3])

    @staticmethod
    def alloc_outputs(N, dev):
        return (torch.empty((N, 4, 4), dtype=torch.float32, device=dev), torch.empty((N, 3), dtype=torch.float32, device=dev),
                torch.empty((N, 3, 3), dtype=torch.float32, device=dev))

    @property
    def last_raw_output(self):
        raw = getattr(self, "_raw_parts", None)
        if not raw or raw[0] is None:
            return None
        return {k: torch.cat([r[k] for r in raw], 0) for k in raw[0]}

    @torch.inference_mode()
    def predict(self, rgb, depth, K, ob_in_cams, xyz_map, normal_map=None, get_vis=False, mesh=None,
                mesh_tensors=None, glctx=None, mesh_diameter=None, iteration=5, shared_translation=None, graph=None):
        """@rgb (H,W,3) uint8/float np or tensor; @ob_in_cams (N,4,4) np or tensor.  -> ((N,4,4) f32 device tensor, vis).
        shared_translation (not in the reference's signature): True = the caller guarantees that all hypotheses have the
        same translation (register()); None = found out here when ob_in_cams is host data (a device tensor is not read
        back: treated as False); False = never share.  It only removes repeated work (refine_part), never changes a bit.
        graph (not in the reference's signature either): None = the predictor's setting (constructor), else True / "auto" /
        False as there -- hipGraph replay of the loop (refine_graphed), bit-identical to the eager launches."""
        self.plan()
        dev = self._plan_dev
        shared_translation = resolve_shared_translation(ob_in_cams, shared_translation)
        if mesh_tensors is None:
            mesh_tensors = make_mesh_tensors(mesh, device=dev)
        B_in_cams = torch.as_tensor(ob_in_cams, device=dev, dtype=torch.float).reshape(-1, 4, 4).contiguous()
        rgb_t = torch.as_tensor(rgb, device=dev).to(torch.float).contiguous()
        xyz_t = torch.as_tensor(xyz_map, device=dev, dtype=torch.float).contiguous()
        H, W = int(rgb_t.shape[0]), int(rgb_t.shape[1])
        done = self.refine_graphed(rgb_t, xyz_t, B_in_cams, K, H, W, mesh_tensors, mesh_diameter, int(iteration),
                                   bool(shared_translation), self.graph if graph is None else graph)
        B_in_cams, trans, rot = done if done is not None else \
            self.refine_device(rgb_t, xyz_t, B_in_cams, K, H, W, get_mesh_handle(mesh_tensors), mesh_diameter, iteration,
                               shared_translation=bool(shared_translation))
        self.last_trans_update = trans
        self.last_rot_update = rot
        if get_vis:
            # debug canvas (predict_pose_refine.py:241-291): network inputs at the initial poses above those at the refined poses
            from .vis import crop_rows_canvas, make_grid_image
            P_in = torch.as_tensor(ob_in_cams, device=dev, dtype=torch.float).reshape(-1, 4, 4).contiguous()
            canv = []
            for P in (P_in, B_in_cams):
                AB = torch.empty((2 * P.shape[0], 6, int(self.cfg["input_resize"][0]), int(self.cfg["input_resize"][1])),
                                 dtype=torch.float32, device=dev)
                b = make_crop_data_batch(self.cfg["input_resize"], P, mesh, rgb_t, None, K, self.cfg["crop_ratio"], xyz_t, cfg=self.cfg,
                                         mesh_tensors=mesh_tensors, mesh_diameter=mesh_diameter, AB=AB, depth_hw=(H, W))
                n = P.shape[0]
                canv.append(crop_rows_canvas(b.AB[:n].cpu().numpy(), b.AB[n:].cpu().numpy()))
            return B_in_cams, make_grid_image(canv, nrow=2, padding=2, pad_value=255)
        return B_in_cams, None
