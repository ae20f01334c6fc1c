"""Mints tests/golden/readers_golden.npz + the small on-disk datasets under tests/golden/reader_seq/ that pin
foundationpose_amd/datareader.py against the REFERENCE's own readers.

The datasets are written with this repo's writers (write_sequence, write_bop_scene) in the two layouts the reference
reads; they are then read back by the reference's `YcbineoatReader` and `BopBaseReader` classes, imported unmodified from
/root/reference/datareader.py (build container only).  cv2 and imageio are not installed here: the handful of calls
the readers make (imread unchanged / colour, nearest-neighbour resize, imageio.imread) are served by PIL + numpy
stand-ins with OpenCV's conventions (BGR channel order for colour imread, nearest resize = floor(dst * src/dst)).
The reference's outputs go into the .npz; tests/test_readers_vs_reference_golden.py reads the committed directories with
OUR readers and compares.

    python tests/golden/make_golden_readers.py
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


class cv2_standin:
    INTER_NEAREST = 0
    INTER_LINEAR = 1

    @staticmethod
    def imread(path, flag=1):
        from PIL import Image
        a = np.array(Image.open(path))
        if flag == -1:                       # IMREAD_UNCHANGED
            return a[..., ::-1].copy() if a.ndim == 3 else a
        if a.ndim == 2:
            a = np.repeat(a[..., None], 3, axis=2)
        return a[..., :3][..., ::-1].astype(np.uint8).copy()

    @staticmethod
    def resize(src, dsize=None, fx=None, fy=None, interpolation=1):
        H, W = src.shape[:2]
        if dsize is None:
            dsize = (int(round(W * fx)), int(round(H * fy)))
        w, h = int(dsize[0]), int(dsize[1])
        if (h, w) == (H, W):
            return src.copy()
        assert interpolation == cv2_standin.INTER_NEAREST, "only the nearest-neighbour resize is stood in for"
        ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1)
        xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
        return src[ys][:, xs].copy()


class imageio_standin:
    @staticmethod
    def imread(path):
        from PIL import Image
        return np.array(Image.open(path))


def make_data(root):
    """three 64x48 frames of a synthetic scene in both layouts"""
    from foundationpose_amd.datareader import write_bop_scene, write_sequence
    rng = np.random.default_rng(5)
    H, W, F = 48, 64, 3
    K = np.array([[60.0, 0, 31.5], [0, 61.0, 23.5], [0, 0, 1]])
    colors = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    depths = rng.uniform(0.3, 1.6, size=(F, H, W))
    depths[:, :4] = 0.0                                  # holes
    depths[:, 5, ::3] = 0.0004                           # below the 1 mm threshold after quantisation
    masks = np.zeros((F, H, W), np.uint8)
    masks[:, 10:30, 20:45] = 1
    masks2 = np.zeros((F, H, W), np.uint8)
    masks2[:, 25:40, 5:25] = 1
    poses = np.tile(np.eye(4), (F, 1, 1))
    for f in range(F):
        a = 0.3 * (f + 1)
        poses[f, :3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        poses[f, :3, 3] = [0.01 * f, -0.02, 0.7 + 0.05 * f]
    shutil.rmtree(root, ignore_errors=True)
    write_sequence(os.path.join(root, "demo"), K, colors, depths, masks, gt_poses=poses)
    inst = [[(5, poses[f], masks[f]), (12, poses[(f + 1) % F], masks2[f])] for f in range(F)]
    write_bop_scene(os.path.join(root, "bop", "000048"), K, colors, depths, inst, depth_scale=0.1)
    return K


def main():
    import ref_harness as rh
    rh.load_reference()
    import datareader as ref_dr            # /root/reference/datareader.py
    assert ref_dr.__file__.startswith("/root/reference"), ref_dr.__file__
    ref_dr.cv2 = cv2_standin
    ref_dr.imageio = imageio_standin
    root = os.path.join(HERE, "reader_seq")
    make_data(root)
    out = {}
    for tag, kw in (("full", dict(zfar=1.2)), ("half", dict(shorter_side=24, zfar=np.inf))):
        r = ref_dr.YcbineoatReader(os.path.join(root, "demo"), **kw)
        out[f"demo_{tag}_K"] = np.asarray(r.K, dtype=np.float64)
        out[f"demo_{tag}_HW"] = np.array([r.H, r.W, len(r)])
        out[f"demo_{tag}_ids"] = np.array(r.id_strs)
        out[f"demo_{tag}_video_name"] = np.array(r.get_video_name())
        for i in range(len(r)):
            out[f"demo_{tag}_color{i}"] = r.get_color(i)
            out[f"demo_{tag}_depth{i}"] = r.get_depth(i)
            out[f"demo_{tag}_mask{i}"] = r.get_mask(i)
            out[f"demo_{tag}_pose{i}"] = r.get_gt_pose(i)
        out[f"demo_{tag}_xyz0"] = np.asarray(r.get_xyz_map(0))
    for tag, kw in (("r1", dict(zfar=1.5, resize=1)),):
        r = ref_dr.BopBaseReader(os.path.join(root, "bop", "000048"), **kw)
        out[f"bop_{tag}_n"] = np.array([len(r.color_files)])
        out[f"bop_{tag}_ids"] = np.array(r.id_strs)
        out[f"bop_{tag}_video_id"] = np.array([r.get_video_dir()])
        out[f"bop_{tag}_depth_scale"] = np.array([r.bop_depth_scale])
        for i in range(len(r.color_files)):
            out[f"bop_{tag}_K{i}"] = np.asarray(r.get_K(i), dtype=np.float64)
            out[f"bop_{tag}_color{i}"] = r.get_color(i)
            out[f"bop_{tag}_depth{i}"] = r.get_depth(i)
            out[f"bop_{tag}_obs{i}"] = np.asarray(r.get_instance_ids_in_image(i))
            for ob in (5, 12):
                out[f"bop_{tag}_mask{i}_{ob}"] = np.asarray(r.get_mask(i, ob))
                out[f"bop_{tag}_pose{i}_{ob}"] = np.asarray(r.get_gt_pose(i, ob))
                out[f"bop_{tag}_poses{i}_{ob}"] = np.asarray(r.get_gt_poses(i, ob))
        out[f"bop_{tag}_xyz0"] = np.asarray(r.get_xyz_map(0))
    # ---- pose-error metrics and projection helper (Utils.py:232-266, :667-672), the reference's functions as they are
    U = rh.load_reference().Utils
    rng = np.random.default_rng(9)
    pts = rng.uniform(-0.05, 0.05, size=(500, 3))
    pts[:, 2] *= 2.0
    out["metric_pts"] = pts
    preds, gts = [], []
    for k in range(6):
        def rand_pose(scale):
            w = rng.normal(size=3) * scale
            th = np.linalg.norm(w)
            Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / max(th, 1e-12)
            T = np.eye(4)
            T[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
            T[:3, 3] = rng.normal(size=3) * 0.02 * scale + [0, 0, 0.6]
            return T
        gts.append(rand_pose(1.0))
        preds.append(gts[-1] @ rand_pose(0.05 * (k + 1)) @ np.diag([1, 1, 1, 1.0]))
        preds[-1][:3, 3] = gts[-1][:3, 3] + rng.normal(size=3) * 0.004 * (k + 1)
    out["metric_pred"], out["metric_gt"] = np.array(preds), np.array(gts)
    out["metric_add"] = np.array([U.add_err(p, g, pts) for p, g in zip(preds, gts)])
    out["metric_adds"] = np.array([U.adds_err(p, g, pts) for p, g in zip(preds, gts)])
    errs = np.abs(rng.normal(size=200)) * 0.04
    out["metric_errs"] = errs
    out["metric_auc"] = np.array([U.compute_auc_sklearn(errs), U.compute_auc_sklearn(errs, max_val=0.05, step=0.0005),
                                  U.compute_auc_sklearn(out["metric_add"]), U.compute_auc_sklearn(np.zeros(5))])
    Kc = np.array([[600.0, 0, 320], [0, 610.0, 240], [0, 0, 1]])
    out["metric_K"] = Kc
    out["metric_proj"] = np.array([U.project_3d_to_2d(np.append(pts[i], 1.0), Kc, gts[i % 6]) for i in range(12)])
    np.savez_compressed(os.path.join(HERE, "readers_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
