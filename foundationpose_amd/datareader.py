"""Demo-format RGB-D sequences: the directory layout the reference's `YcbineoatReader`
(datareader.py:57-152) reads -- `rgb/*.png`, `depth/*.png` (uint16 millimetres), `masks/*.png`, `cam_K.txt`,
optional `annotated_poses/*.txt` -- with PIL + numpy instead of cv2 / imageio.  Same attribute and method names, same
value conventions (depth in metres with values < 0.001 or >= zfar zeroed, masks as uint8 0/1, nearest resize)."""
import glob
import logging
import os

import numpy as np


def _imread(path):
    from PIL import Image
    return np.array(Image.open(path))   # a writable copy (torch.as_tensor warns on PIL's read-only buffer)


def _resize_nearest(a, W, H):
    if a.shape[0] == H and a.shape[1] == W:
        return a
    ys = np.minimum((np.arange(H) * (a.shape[0] / H)).astype(np.int64), a.shape[0] - 1)   # cv2.INTER_NEAREST: floor(dst*scale)
    xs = np.minimum((np.arange(W) * (a.shape[1] / W)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


class YcbineoatReader:
    def __init__(self, video_dir, downscale=1, shorter_side=None, zfar=np.inf):
        self.video_dir = video_dir
        self.downscale = downscale
        self.zfar = zfar
        self.color_files = sorted(glob.glob(f"{self.video_dir}/rgb/*.png"))
        if not self.color_files:
            raise FileNotFoundError(f"no frames under {self.video_dir}/rgb/*.png")
        self.K = np.loadtxt(f"{video_dir}/cam_K.txt").reshape(3, 3)
        self.id_strs = [os.path.basename(f).replace(".png", "") for f in self.color_files]
        self.H, self.W = _imread(self.color_files[0]).shape[:2]
        if shorter_side is not None:
            self.downscale = shorter_side / min(self.H, self.W)
        self.H = int(self.H * self.downscale)
        self.W = int(self.W * self.downscale)
        self.K[:2] *= self.downscale
        self.gt_pose_files = sorted(glob.glob(f"{self.video_dir}/annotated_poses/*"))

    def get_video_name(self):
        return self.video_dir.rstrip("/").split("/")[-1]

    def __len__(self):
        return len(self.color_files)

    def get_gt_pose(self, i):
        try:
            return np.loadtxt(self.gt_pose_files[i]).reshape(4, 4)
        except Exception:
            logging.info("GT pose not found, return None")
            return None

    def get_color(self, i):
        return np.ascontiguousarray(_resize_nearest(_imread(self.color_files[i])[..., :3], self.W, self.H))

    def get_mask(self, i):
        mask = _imread(self.color_files[i].replace("rgb", "masks"))
        if mask.ndim == 3:
            for c in range(3):
                if mask[..., c].sum() > 0:
                    mask = mask[..., c]
                    break
            else:
                mask = mask[..., 0]
        return _resize_nearest(mask, self.W, self.H).astype(bool).astype(np.uint8)

    def get_depth(self, i):
        depth = _imread(self.color_files[i].replace("rgb", "depth")).astype(np.float64) / 1e3
        depth = _resize_nearest(depth, self.W, self.H).copy()
        depth[(depth < 0.001) | (depth >= self.zfar)] = 0
        return depth

    def get_xyz_map(self, i):
        from .Utils import depth2xyzmap
        return depth2xyzmap(self.get_depth(i), self.K)


def write_sequence(video_dir, K, colors, depths, masks, gt_poses=None):
    """writes frames in the layout above (depth quantised to uint16 millimetres, as the cameras deliver it)"""
    from PIL import Image
    for sub in ("rgb", "depth", "masks") + (("annotated_poses",) if gt_poses is not None else ()):
        os.makedirs(os.path.join(video_dir, sub), exist_ok=True)
    np.savetxt(os.path.join(video_dir, "cam_K.txt"), np.asarray(K, dtype=np.float64).reshape(3, 3))
    for i, (c, d, m) in enumerate(zip(colors, depths, masks)):
        name = f"{i:07d}"
        Image.fromarray(np.asarray(c, dtype=np.uint8)).save(os.path.join(video_dir, "rgb", name + ".png"))
        mm = np.clip(np.rint(np.asarray(d, dtype=np.float64) * 1e3), 0, 65535).astype(np.uint16)
        Image.fromarray(mm).save(os.path.join(video_dir, "depth", name + ".png"))
        Image.fromarray((np.asarray(m) > 0).astype(np.uint8) * 255).save(os.path.join(video_dir, "masks", name + ".png"))
        if gt_poses is not None:
            np.savetxt(os.path.join(video_dir, "annotated_poses", name + ".txt"), np.asarray(gt_poses[i]).reshape(4, 4))


class BopBaseReader:
    """One BOP scene directory (reference datareader.py:155-365): `rgb/` (or `gray/`) + `depth/` (uint16, value *
    depth_scale = millimetres) + `mask_visib/<frame>_<instance>.png` + `scene_camera.json` + `scene_gt.json`; object
    models as `<models_dir>/obj_<id>.ply|.obj` in millimetres with `models_info.json` beside them.  Poses come back in
    metres, ob_in_cam, 4x4 float64."""

    def __init__(self, base_dir, zfar=np.inf, resize=1, models_dir=None):
        import json
        self.base_dir = base_dir.rstrip("/")
        self.resize = resize
        self.zfar = zfar
        self.dataset_name = None
        self.models_dir = models_dir
        self.color_files = sorted(glob.glob(f"{self.base_dir}/rgb/*")) or sorted(glob.glob(f"{self.base_dir}/gray/*"))
        if not self.color_files:
            raise FileNotFoundError(f"no frames under {self.base_dir}/rgb or {self.base_dir}/gray")
        with open(f"{self.base_dir}/scene_camera.json") as f:
            cam = json.load(f)
        self.K_table, self.depth_scale_table = {}, {}
        for k, v in cam.items():
            self.K_table[f"{int(k):06d}"] = np.array(v["cam_K"], dtype=np.float64).reshape(3, 3)
            self.depth_scale_table[f"{int(k):06d}"] = float(v.get("depth_scale", 1.0))
        self.bop_depth_scale = next(iter(self.depth_scale_table.values()))
        gt_file = f"{self.base_dir}/scene_gt.json"
        self.scene_gt = None
        if os.path.exists(gt_file):
            with open(gt_file) as f:
                self.scene_gt = json.load(f)
            assert len(self.scene_gt) == len(self.color_files), "scene_gt.json does not cover every frame"
        self.id_strs = [os.path.basename(f).split(".")[0] for f in self.color_files]
        self.scene_ob_ids_dict = None

    def __len__(self):
        return len(self.color_files)

    def get_video_id(self):
        return int(self.base_dir.split("/")[-1])

    def get_K(self, i_frame):
        K = self.K_table[self.id_strs[i_frame]].copy()
        if self.resize != 1:
            K[:2, :2] *= self.resize     # the reference scales fx, fy (and the skew) only, datareader.py:199-203
        return K

    def _frame_gt(self, i_frame):
        return self.scene_gt[str(int(self.id_strs[i_frame]))]

    def get_instance_ids_in_image(self, i_frame):
        if self.scene_gt is not None:
            return np.asarray([k["obj_id"] for k in self._frame_gt(i_frame)])
        if self.scene_ob_ids_dict is not None:
            return np.asarray(self.scene_ob_ids_dict[self.id_strs[i_frame]])
        mask_dir = os.path.dirname(self.color_files[0]).replace("rgb", "mask_visib")
        files = sorted(glob.glob(f"{mask_dir}/{self.id_strs[i_frame]}_*.png"))
        return np.asarray([int(os.path.basename(f).split(".")[0].split("_")[1]) for f in files])

    def _resized(self, a):
        if self.resize == 1:
            return a
        return _resize_nearest(a, int(round(a.shape[1] * self.resize)), int(round(a.shape[0] * self.resize)))

    def get_color(self, i):
        color = _imread(self.color_files[i])
        if color.ndim == 2:
            color = np.tile(color[..., None], (1, 1, 3))
        return np.ascontiguousarray(self._resized(color[..., :3]))

    def get_depth(self, i, filled=False):
        f = self.color_files[i].replace("/rgb/", "/depth/").replace("/gray/", "/depth/")
        if not f.endswith(".png"):
            f = os.path.splitext(f)[0] + ".png"
        depth = _imread(f).astype(np.float64) * 1e-3 * self.depth_scale_table[self.id_strs[i]]
        depth = self._resized(depth).copy()
        depth[(depth < 0.001) | (depth > self.zfar)] = 0
        return depth

    def get_xyz_map(self, i):
        from .Utils import depth2xyzmap
        return depth2xyzmap(self.get_depth(i), self.get_K(i))

    def get_mask(self, i_frame, ob_id, type="mask_visib"):
        """mask of the FIRST instance of ob_id in the frame, as a bool array (None if the file is missing)"""
        if self.scene_gt is None:
            raise RuntimeError("get_mask needs scene_gt.json")
        pos = 0
        for k in self._frame_gt(i_frame):
            if k["obj_id"] == ob_id:
                break
            pos += 1
        f = f"{self.base_dir}/{type}/{int(self.id_strs[i_frame]):06d}_{pos:06d}.png"
        if not os.path.exists(f):
            logging.info(f"{f} not found")
            return None
        return self._resized(_imread(f)) > 0

    @staticmethod
    def _pose(k):
        T = np.eye(4)
        T[:3, :3] = np.array(k["cam_R_m2c"], dtype=np.float64).reshape(3, 3)
        T[:3, 3] = np.array(k["cam_t_m2c"], dtype=np.float64) / 1e3
        return T

    def get_gt_poses(self, i_frame, ob_id):
        return np.asarray([self._pose(k) for k in self._frame_gt(i_frame) if k["obj_id"] == ob_id]).reshape(-1, 4, 4)

    def get_gt_pose(self, i_frame, ob_id, mask=None, use_my_correction=False):
        """with several instances of ob_id, `mask` picks the one whose visible mask overlaps it most (IoU)"""
        best, best_iou = np.eye(4), -np.inf
        for i_k, k in enumerate(self._frame_gt(i_frame)):
            if k["obj_id"] != ob_id:
                continue
            if mask is None:
                return self._pose(k)
            gt_mask = _imread(f"{self.base_dir}/mask_visib/{self.id_strs[i_frame]}_{i_k:06d}.png") > 0
            union = np.logical_or(gt_mask, mask).sum()
            iou = float(np.logical_and(gt_mask, mask).sum()) / union if union else 0.0
            if iou > best_iou:
                best, best_iou = self._pose(k), iou
        return best

    # ---- object models
    def get_gt_mesh_file(self, ob_id):
        if self.models_dir is None:
            raise RuntimeError("You should override this")
        for ext in (".obj", ".ply"):
            f = f"{self.models_dir}/obj_{int(ob_id):06d}{ext}"
            if os.path.exists(f):
                return f
        raise FileNotFoundError(f"no model for object {ob_id} under {self.models_dir}")

    def get_gt_mesh(self, ob_id):
        from .mesh_io import load_mesh
        mesh = load_mesh(self.get_gt_mesh_file(ob_id))
        mesh.vertices = np.asarray(mesh.vertices, dtype=np.float64) * 1e-3     # BOP models are in millimetres
        return mesh

    def _models_info(self):
        import json
        with open(f"{os.path.dirname(self.get_gt_mesh_file(self.ob_ids[0]))}/models_info.json") as f:
            return json.load(f)

    def get_model_diameter(self, ob_id):
        return self._models_info()[str(ob_id)]["diameter"] / 1e3

    def load_symmetry_tfs(self):
        import copy
        from .Utils import symmetry_tfs_from_info
        info = self._models_info()
        self.symmetry_tfs, self.symmetry_info_table = {}, {}
        for ob_id in self.ob_ids:
            self.symmetry_info_table[ob_id] = info[str(ob_id)]
            self.symmetry_tfs[ob_id] = symmetry_tfs_from_info(info[str(ob_id)], rot_angle_discrete=5)
        self.geometry_symmetry_info_table = copy.deepcopy(self.symmetry_info_table)


class YcbVideoReader(BopBaseReader):
    """YCB-Video in BOP layout (reference datareader.py:433-530): 21 objects, models under `$YCB_VIDEO_DIR/models`
    (or `models_dir`), keyframes from `<scene>/../../keyframe.txt` when the scene is not a BOP test split."""

    def __init__(self, base_dir, zfar=np.inf, models_dir=None):
        if models_dir is None and os.getenv("YCB_VIDEO_DIR"):
            models_dir = os.path.join(os.getenv("YCB_VIDEO_DIR"), "models")
        super().__init__(base_dir, zfar=zfar, models_dir=models_dir)
        self.dataset_name = "ycbv"
        self.K = next(iter(self.K_table.values()))
        self.ob_ids = list(range(1, 22))
        self.keyframe_lines = None
        kf = os.path.join(self.base_dir, "..", "..", "keyframe.txt")
        if "BOP" not in self.base_dir and os.path.exists(kf):
            with open(kf) as f:
                self.keyframe_lines = f.read().splitlines()
        present = [o for o in self.ob_ids if self.models_dir and any(
            os.path.exists(f"{self.models_dir}/obj_{o:06d}{e}") for e in (".obj", ".ply"))]
        if present:
            self.ob_ids = present
            self.load_symmetry_tfs()

    def is_keyframe(self, i):
        if self.keyframe_lines is None:
            return True
        return f"{self.get_video_id():04d}/{int(self.id_strs[i]):06d}" in self.keyframe_lines


def write_bop_scene(scene_dir, K, colors, depths, instances, models_dir=None, meshes=None, depth_scale=0.1):
    """Mints a scene in the BOP layout.  instances[i] = list of (obj_id, ob_in_cam 4x4 in metres, visible mask) for
    frame i; meshes = {obj_id: mesh in metres} are written as OBJ in millimetres with a models_info.json."""
    import json
    from PIL import Image
    for sub in ("rgb", "depth", "mask_visib"):
        os.makedirs(os.path.join(scene_dir, sub), exist_ok=True)
    cam, gt = {}, {}
    for i, (c, d, inst) in enumerate(zip(colors, depths, instances)):
        name = f"{i:06d}"
        Image.fromarray(np.asarray(c, dtype=np.uint8)).save(os.path.join(scene_dir, "rgb", name + ".png"))
        units = np.clip(np.rint(np.asarray(d, dtype=np.float64) * 1e3 / depth_scale), 0, 65535).astype(np.uint16)
        Image.fromarray(units).save(os.path.join(scene_dir, "depth", name + ".png"))
        cam[str(i)] = {"cam_K": np.asarray(K, dtype=np.float64).reshape(-1).tolist(), "depth_scale": depth_scale}
        gt[str(i)] = []
        for j, (ob_id, T, m) in enumerate(inst):
            T = np.asarray(T, dtype=np.float64)
            gt[str(i)].append({"obj_id": int(ob_id), "cam_R_m2c": T[:3, :3].reshape(-1).tolist(), "cam_t_m2c": (T[:3, 3] * 1e3).tolist()})
            Image.fromarray((np.asarray(m) > 0).astype(np.uint8) * 255).save(os.path.join(scene_dir, "mask_visib", f"{name}_{j:06d}.png"))
    with open(os.path.join(scene_dir, "scene_camera.json"), "w") as f:
        json.dump(cam, f)
    with open(os.path.join(scene_dir, "scene_gt.json"), "w") as f:
        json.dump(gt, f)
    if models_dir is not None and meshes:
        import copy
        from .mesh_io import save_obj
        from .Utils import compute_mesh_diameter
        os.makedirs(models_dir, exist_ok=True)
        info = {}
        for ob_id, mesh in meshes.items():
            mm = copy.deepcopy(mesh)
            mm.vertices = np.asarray(mesh.vertices, dtype=np.float64) * 1e3
            save_obj(mm, os.path.join(models_dir, f"obj_{int(ob_id):06d}.obj"))
            info[str(ob_id)] = {"diameter": float(compute_mesh_diameter(model_pts=np.asarray(mesh.vertices), n_sample=10000) * 1e3)}
        with open(os.path.join(models_dir, "models_info.json"), "w") as f:
            json.dump(info, f)
