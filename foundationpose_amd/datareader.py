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
