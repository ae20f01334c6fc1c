"""Developer-facing helpers off the hot path: accuracy metrics and debug canvases, re-expressed
with numpy / scipy / PIL (the reference uses cv2, torchvision, sklearn).  Citations are to /root/reference/Utils.py.
Pixel-exact equality with OpenCV's anti-aliased drawing or its JET colour map is not a goal; geometry and value
conventions are the reference's."""
import numpy as np


def _transform_pts(pts, tf):
    return (np.asarray(tf)[:3, :3] @ np.asarray(pts).T).T + np.asarray(tf)[:3, 3]


def add_err(pred, gt, model_pts, symetry_tfs=None):
    """Average Distance of Model Points (Utils.py:232-240)"""
    return float(np.linalg.norm(_transform_pts(model_pts, pred) - _transform_pts(model_pts, gt), axis=-1).mean())


def adds_err(pred, gt, model_pts):
    """ADD-S: mean closest-point distance (Utils.py:242-253)"""
    from scipy.spatial import cKDTree
    nn_dists, _ = cKDTree(_transform_pts(model_pts, pred)).query(_transform_pts(model_pts, gt), k=1)
    return float(nn_dists.mean())


def compute_auc(errs, max_val=0.1, step=0.001):
    """area under the accuracy-threshold curve, normalised to [0,1] (Utils.py:255-266, trapezoid rule like sklearn.metrics.auc)"""
    errs = np.sort(np.asarray(errs, dtype=np.float64))
    X = np.arange(0, max_val + step, step)
    Y = np.ones(len(X))
    for i, x in enumerate(X):
        Y[i] = (errs <= x).sum() / len(errs)
        if Y[i] >= 1:
            break
    return float(np.trapezoid(Y, X) / max_val)


def _jet(v):
    """v in [0,1] -> uint8 RGB, the piecewise-linear JET map"""
    v = np.clip(v, 0, 1)
    r = np.clip(1.5 - np.abs(4 * v - 3), 0, 1)
    g = np.clip(1.5 - np.abs(4 * v - 2), 0, 1)
    b = np.clip(1.5 - np.abs(4 * v - 1), 0, 1)
    return (np.stack([r, g, b], -1) * 255).astype(np.uint8)


def depth_to_vis(depth, zmin=None, zmax=None, mode="rgb", inverse=True):
    """Utils.py:456-480"""
    depth = np.asarray(depth, dtype=np.float64)
    zmin = depth.min() if zmin is None else zmin
    zmax = depth.max() if zmax is None else zmax
    if inverse:
        invalid = depth < 0.001
        vis = zmin / (depth + 1e-8)
        vis[invalid] = 0
    else:
        d = depth.clip(zmin, zmax)
        invalid = (d == zmin) | (d == zmax)
        vis = (d - zmin) / max(zmax - zmin, 1e-12)
        vis[invalid] = 1
    if mode == "gray":
        return (vis * 255).clip(0, 255).astype(np.uint8)
    if mode == "rgb":
        return _jet((vis * 255).astype(np.uint8) / 255.0)
    raise RuntimeError(mode)


def make_grid_image(imgs, nrow, padding=5, pad_value=255):
    """torchvision.utils.make_grid semantics (Utils.py:293-301): imgs (B,H,W,C), nrow images per row"""
    imgs = [np.asarray(i) for i in imgs]
    H = max(i.shape[0] for i in imgs)
    W = max(i.shape[1] for i in imgs)
    n = len(imgs)
    cols = min(nrow, n)
    rows = int(np.ceil(n / cols))
    grid = np.full((rows * (H + padding) + padding, cols * (W + padding) + padding, 3), pad_value, dtype=np.float64)
    for k, im in enumerate(imgs):
        r, c = divmod(k, cols)
        y, x = padding + r * (H + padding), padding + c * (W + padding)
        grid[y:y + im.shape[0], x:x + im.shape[1]] = im[..., :3]
    return grid.clip(0, 255).astype(np.uint8)


def project_3d_to_2d(pt, K, ob_in_cam):
    """Utils.py:666-672"""
    p = np.asarray(K) @ (np.asarray(ob_in_cam) @ np.asarray(pt, dtype=np.float64).reshape(4, 1))[:3]
    p = p.reshape(-1) / p.reshape(-1)[2]
    return p[:2].round().astype(int)


def draw_xyz_axis(color, ob_in_cam, scale=0.1, K=np.eye(3), thickness=3, transparency=0, is_input_rgb=True):
    """object axes x/y/z in red/green/blue on an RGB image (Utils.py:675-710)"""
    from PIL import Image, ImageDraw
    img = Image.fromarray(np.asarray(color, dtype=np.uint8).copy())
    d = ImageDraw.Draw(img)
    o = tuple(int(v) for v in project_3d_to_2d(np.array([0, 0, 0, 1.0]), K, ob_in_cam))
    for axis, col in ((0, (255, 0, 0)), (1, (0, 255, 0)), (2, (0, 0, 255))):
        e = np.array([0, 0, 0, 1.0])
        e[axis] = scale
        d.line([o, tuple(int(v) for v in project_3d_to_2d(e, K, ob_in_cam))], fill=col, width=thickness)
    out = np.asarray(img).astype(np.float64)
    if transparency > 0:
        base = np.asarray(color, dtype=np.float64)
        m = np.linalg.norm(out - base, axis=-1) > 0
        out[m] = base[m] * transparency + out[m] * (1 - transparency)
    return out.astype(np.uint8)


def draw_posed_3d_box(K, img, ob_in_cam, bbox, line_color=(0, 255, 0), linewidth=2):
    """the 12 edges of the object's bounding box (Utils.py:713-749); bbox (2,3) = min / max corner"""
    from PIL import Image, ImageDraw
    bbox = np.asarray(bbox, dtype=np.float64)
    lo, hi = bbox.min(0), bbox.max(0)
    im = Image.fromarray(np.asarray(img, dtype=np.uint8).copy())
    d = ImageDraw.Draw(im)
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    cam = _transform_pts(corners, ob_in_cam)
    uv = (np.asarray(K) @ cam.T).T
    uv = np.round(uv[:, :2] / uv[:, 2:3]).astype(int)
    for a in range(8):
        for b in range(a + 1, 8):
            if bin(a ^ b).count("1") == 1:      # corners that differ along exactly one axis
                d.line([tuple(uv[a]), tuple(uv[b])], fill=tuple(line_color), width=linewidth)
    return np.asarray(im)


def crop_rows_canvas(A, B, ids=None, padding=2, texts=None):
    """debug canvas of network inputs (predict_pose_refine.py:241-262, predict_score.py:27-52): one row per hypothesis
    [rendered rgb | observed rgb | rendered z | observed z]; A, B (N,6,h,w) arrays in network units."""
    A, B = np.asarray(A, dtype=np.float32), np.asarray(B, dtype=np.float32)
    ids = range(A.shape[0]) if ids is None else ids
    rows = []
    for i in ids:
        rgbA = (A[i, :3] * 255).transpose(1, 2, 0)
        rgbB = (B[i, :3] * 255).transpose(1, 2, 0)
        zA, zB = A[i, 5], B[i, 5]
        zmin, zmax = min(zA.min(), zB.min()), max(zA.max(), zB.max())
        rows.append(make_grid_image([rgbA, rgbB, depth_to_vis(zA, zmin, zmax, inverse=False), depth_to_vis(zB, zmin, zmax, inverse=False)],
                                    nrow=4, padding=padding, pad_value=255))
    return make_grid_image(rows, nrow=1, padding=padding, pad_value=255)
