"""Runs the REFERENCE's own Python for the hot path on CPU, in the build container only (/root/reference is not on
the GPU box).  TEST INFRASTRUCTURE: used by tests/golden/make_golden_pipeline.py to mint the golden vectors that pin
oracle/ (committed under tests/golden/*.npz).

What executes is the reference's code, unmodified and imported from /root/reference:
  Utils.py (compute_crop_window_tf_batch, transform_pts, nvdiffrast_render, projection_matrix_from_intrinsics,
  depth2xyzmap[_batch], egocentric_delta_pose_to_pose, ...), learning/training/predict_pose_refine.py
  (make_crop_data_batch, PoseRefinePredictor.predict), learning/training/predict_score.py (make_crop_data_batch,
  ScorePredictor.predict), learning/datasets/{h5_dataset,pose_dataset}.py (transform_batch, BatchPoseData),
  learning/models/*.py (RefineNet, ScoreNetMultiPair).

What does NOT exist in this container and is replaced by the small stand-ins below, written from the packages'
published semantics -- these are the only "unpinned" pieces left:
  nvdiffrast.torch  : rasterize / interpolate / texture           (App. B.1)
  kornia            : geometry.transform.warp_perspective          (App. B.2, on top of torch's F.grid_sample)
  pytorch3d         : so3_exp_map, rotation_6d_to_matrix           (App. B.3)
  trimesh           : creation.icosphere (vertices only)           (App. B.5; trimesh 4.2.2 creation.icosahedron + remesh.subdivide)
  transformations   : euler_matrix (static xyz), Gohlke's formula restated, cross-checked against scipy's Rotation
  mycpp             : cluster_poses -- float32 numpy restatement of mycpp/src/app/pybind_api.cpp:24-68 + Utils.cpp:21-26
                      (Eigen / Boost are not in this image, so the C++ cannot be compiled)
Every other missing import (trimesh, open3d, cv2, warp, omegaconf, ...) is a MagicMock: imported, never called on
this path.  'cuda' device requests are redirected to the CPU.
"""
import importlib
import math
import sys
import types
from unittest import mock

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"


# ----------------------------------------------------------------------------------------------- nvdiffrast stand-in
class _Ctx:
    pass


def _rasterize(glctx, pos, tri, resolution, **kw):
    """dr.rasterize in instanced mode (App. B.1): pos (N,V,4) clip space f32, tri (T,3) i32, resolution (H,W).
    -> (N,H,W,4) = (u, v, z/w, tri_id+1), row 0 = bottom row.  Coverage: vertices snapped to 1/16 px, integer edge
    functions, top-left rule, nearest depth wins (integer key, lower triangle index on ties) -- the same definition as
    SURVEY App. A.8; barycentrics u,v and z/w recomputed per pixel from the UNSNAPPED clip-space vertices."""
    pos = pos.detach().cpu().numpy().astype(np.float32)
    tri = tri.detach().cpu().numpy().astype(np.int64)
    H, W = int(resolution[0]), int(resolution[1])
    N = pos.shape[0]
    out = np.zeros((N, H, W, 4), np.float32)
    for n in range(N):
        p = pos[n]
        w = p[:, 3]
        ok = w > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            xw = (p[:, 0] / w * np.float32(0.5) + np.float32(0.5)) * np.float32(W)
            yw = (p[:, 1] / w * np.float32(0.5) + np.float32(0.5)) * np.float32(H)
        xs = np.rint(xw.astype(np.float64) * 16)
        ys = np.rint(yw.astype(np.float64) * 16)
        ok &= np.isfinite(xs) & np.isfinite(ys) & (np.abs(xs) < 1 << 20) & (np.abs(ys) < 1 << 20)
        xs = np.where(ok, xs, 0).astype(np.int64)
        ys = np.where(ok, ys, 0).astype(np.int64)
        best_key = np.full((H, W), np.iinfo(np.int64).max, np.int64)
        for t in range(tri.shape[0]):
            i0, i1, i2 = tri[t]
            if not (ok[i0] and ok[i1] and ok[i2]):
                continue
            x0, y0, x1, y1, x2, y2 = xs[i0], ys[i0], xs[i1], ys[i1], xs[i2], ys[i2]
            area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
            if area == 0:
                continue
            w1i, w2i = w[i1], w[i2]
            if area < 0:
                x1, y1, x2, y2 = x2, y2, x1, y1
                w1i, w2i = w2i, w1i
                area = -area
            px0 = max(0, int(-(-(min(x0, x1, x2) - 8) // 16)))
            px1 = min(W - 1, int((max(x0, x1, x2) - 8) // 16))
            py0 = max(0, int(-(-(min(y0, y1, y2) - 8) // 16)))
            py1 = min(H - 1, int((max(y0, y1, y2) - 8) // 16))
            if px0 > px1 or py0 > py1:
                continue
            PX, PY = np.meshgrid(np.arange(px0, px1 + 1) * 16 + 8, np.arange(py0, py1 + 1) * 16 + 8)

            def own(dx, dy):
                return (dy > 0) or (dy == 0 and dx < 0)
            e0 = (x2 - x1) * (PY - y1) - (y2 - y1) * (PX - x1)
            e1 = (x0 - x2) * (PY - y2) - (y0 - y2) * (PX - x2)
            e2 = (x1 - x0) * (PY - y0) - (y1 - y0) * (PX - x0)
            inside = ((e0 > 0) | ((e0 == 0) & own(x2 - x1, y2 - y1))) & ((e1 > 0) | ((e1 == 0) & own(x0 - x2, y0 - y2))) & \
                     ((e2 > 0) | ((e2 == 0) & own(x1 - x0, y1 - y0)))
            if not inside.any():
                continue
            # camera depth (w_clip = z_cam for the reference's projection) from the snapped-edge weights, f32
            S = (e0.astype(np.float32) * (np.float32(1) / w[i0])) + e1.astype(np.float32) * (np.float32(1) / w1i) \
                + e2.astype(np.float32) * (np.float32(1) / w2i)
            with np.errstate(divide="ignore", invalid="ignore"):
                zc = np.float32(area) / S
            zq = np.rint(np.minimum(zc, np.float32(4095.0)).astype(np.float64) * 1048576.0).astype(np.int64)
            key = (zq << 32) | t
            sub = best_key[py0:py1 + 1, px0:px1 + 1]
            upd = inside & (key < sub)
            sub[upd] = key[upd]
        cov = best_key != np.iinfo(np.int64).max
        tid = (best_key & 0xFFFFFFFF).astype(np.int64)
        jj, ii = np.nonzero(cov)
        if len(jj):
            t = tid[jj, ii]
            v = [p[tri[t, k]].astype(np.float64) for k in range(3)]
            fx = (2 * ii + 1) / W - 1.0
            fy = (2 * jj + 1) / H - 1.0
            q = [(vk[:, 0] - fx * vk[:, 3], vk[:, 1] - fy * vk[:, 3]) for vk in v]
            a0 = q[1][0] * q[2][1] - q[1][1] * q[2][0]
            a1 = q[2][0] * q[0][1] - q[2][1] * q[0][0]
            a2 = q[0][0] * q[1][1] - q[0][1] * q[1][0]
            iw = 1.0 / (a0 + a1 + a2)
            out[n, jj, ii, 0] = np.clip(a0 * iw, 0, 1)
            out[n, jj, ii, 1] = np.clip(a1 * iw, 0, 1)
            zw = (v[0][:, 2] * a0 + v[1][:, 2] * a1 + v[2][:, 2] * a2) / (v[0][:, 3] * a0 + v[1][:, 3] * a1 + v[2][:, 3] * a2)
            out[n, jj, ii, 2] = np.clip(zw, -1, 1)
            out[n, jj, ii, 3] = t + 1
    return torch.from_numpy(out), None


def _interpolate(attr, rast, tri, **kw):
    """dr.interpolate: sum_k b_k attr[tri[id,k]], zeros where empty; attr (N,V,C) or (V,C)."""
    tri = tri.long()
    N, H, W, _ = rast.shape
    tid = rast[..., 3].long() - 1
    valid = tid >= 0
    idx = tri[tid.clamp(min=0)]                      # (N,H,W,3)
    if attr.dim() == 2:
        a = attr[idx]                                # (N,H,W,3,C)
    else:
        a = torch.stack([attr[n][idx[n]] for n in range(N)], 0)
    u, v = rast[..., 0:1], rast[..., 1:2]
    out = a[..., 0, :] * u + a[..., 1, :] * v + a[..., 2, :] * (1 - u - v)
    return out * valid[..., None].to(out.dtype), None


def _texture(tex, uv, filter_mode="linear", boundary_mode="wrap", **kw):
    """dr.texture(filter_mode='linear', boundary 'wrap'): texel centres at (i+0.5)/W, row 0 = v 0."""
    assert filter_mode == "linear" and tex.shape[0] == 1
    T = tex[0]
    Ht, Wt = T.shape[0], T.shape[1]
    x = uv[..., 0] * Wt - 0.5
    y = uv[..., 1] * Ht - 0.5
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = (x - x0)[..., None], (y - y0)[..., None]
    x0i, y0i = x0.long() % Wt, y0.long() % Ht
    x1i, y1i = (x0i + 1) % Wt, (y0i + 1) % Ht
    a = T[y0i, x0i] * (1 - fx) + T[y0i, x1i] * fx
    b = T[y1i, x0i] * (1 - fx) + T[y1i, x1i] * fx
    return a * (1 - fy) + b * fy


def _make_dr():
    m = types.ModuleType("nvdiffrast.torch")
    m.RasterizeCudaContext = lambda *a, **k: _Ctx()
    m.RasterizeGLContext = lambda *a, **k: _Ctx()
    m.rasterize, m.interpolate, m.texture = _rasterize, _interpolate, _texture
    return m


# ----------------------------------------------------------------------------------------------- kornia stand-in
def _normal_transform_pixel(h, w):
    return torch.tensor([[2.0 / (w - 1), 0, -1.0], [0, 2.0 / (h - 1), -1.0], [0, 0, 1.0]], dtype=torch.float64)


def _warp_perspective(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=True):
    """kornia 0.7.2 geometry.transform.warp_perspective (App. B.2): normalise the homography with the
    [0,w-1]->[-1,1] pixel normalisation, invert, transform a linspace(-1,1) meshgrid, F.grid_sample."""
    B, C, H, W = src.shape
    h_out, w_out = int(dsize[0]), int(dsize[1])
    Ns = _normal_transform_pixel(H, W).to(M.dtype)
    Nd = _normal_transform_pixel(h_out, w_out).to(M.dtype)
    dst_norm_trans_src_norm = Nd[None] @ (M @ torch.linalg.inv(Ns)[None])
    src_norm_trans_dst_norm = torch.linalg.inv(dst_norm_trans_src_norm)
    xs = torch.linspace(-1, 1, w_out, dtype=M.dtype)
    ys = torch.linspace(-1, 1, h_out, dtype=M.dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    grid = torch.stack([gx, gy, torch.ones_like(gx)], -1).reshape(1, -1, 3)          # (1,hw,3)
    g = grid @ src_norm_trans_dst_norm.transpose(1, 2)                               # (B,hw,3)
    g = g[..., :2] / g[..., 2:3]
    return F.grid_sample(src, g.reshape(B, h_out, w_out, 2).to(src.dtype), mode=mode, padding_mode=padding_mode,
                         align_corners=align_corners)


def _make_kornia():
    k = types.ModuleType("kornia")
    k.geometry = types.ModuleType("kornia.geometry")
    k.geometry.transform = types.ModuleType("kornia.geometry.transform")
    k.geometry.transform.warp_perspective = _warp_perspective
    return k


# ----------------------------------------------------------------------------------------------- pytorch3d stand-in
def _so3_exp_map(log_rot, eps=1e-4):
    nrms = (log_rot * log_rot).sum(1)
    theta = torch.clamp(nrms, eps).sqrt()
    fac1 = theta.sin() / theta
    fac2 = (1 - theta.cos()) / (theta * theta)
    x, y, z = log_rot[:, 0], log_rot[:, 1], log_rot[:, 2]
    zero = torch.zeros_like(x)
    Kx = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], 1).reshape(-1, 3, 3)
    return fac1[:, None, None] * Kx + fac2[:, None, None] * (Kx @ Kx) + torch.eye(3, dtype=log_rot.dtype)[None]


def _rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


# ----------------------------------------------------------------------------------------------- trimesh stand-in
def _icosphere(subdivisions=3, radius=1.0, **kw):
    """trimesh.creation.icosphere [3P]: icosahedron() scaled to the unit sphere, then per level remesh.subdivide (one
    midpoint per unique edge; new vertices appended in the order of grouping.unique_rows' integer row hash of the sorted
    edge, low index in the low 32 bits) followed by the re-projection v += v/|v| * (radius - |v|).  Written with dicts
    and explicit sorting, independently of foundationpose_amd.Utils.icosphere_vertices."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = [np.array(p, dtype=np.float64) / np.sqrt(2.0 + t) for p in
             ([-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1],
              [t, 0, 1], [-t, 0, -1], [-t, 0, 1])]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(int(subdivisions)):
        edges = set()
        for a, b, c in faces:
            for e in ((a, b), (b, c), (c, a)):
                edges.add((min(e), max(e)))
        order = sorted(edges, key=lambda e: (e[1], e[0]))           # hash = lo | hi << 32
        index = {e: len(verts) + i for i, e in enumerate(order)}
        new = [(verts[lo] + verts[hi]) / 2.0 for lo, hi in order]
        nf = []
        for a, b, c in faces:
            ab, bc, ca = index[(min(a, b), max(a, b))], index[(min(b, c), max(b, c))], index[(min(c, a), max(c, a))]
            nf += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        faces = nf
        verts = verts + new
        out = []
        for p in verts:
            scalar = np.sqrt(p[0] ** 2 + p[1] ** 2 + p[2] ** 2)
            out.append(p + (p / scalar) * (radius - scalar))
        verts = out
    return types.SimpleNamespace(vertices=np.array(verts), faces=np.array(faces))


def _make_transformations():
    """transformations.euler_matrix for the default axes 'sxyz' [3P: Gohlke's transformations.py, restated]: with
    i,j,k = 0,1,2 and no parity / repetition / frame flip it fills M[i,i]=cj*ck, M[i,j]=sj*sc-cs, M[i,k]=sj*cc+ss,
    M[j,i]=cj*sk, M[j,j]=sj*ss+cc, M[j,k]=sj*cs-sc, M[k,i]=-sj, M[k,j]=cj*si, M[k,k]=cj*ci (cc=ci*ck, cs=ci*sk, sc=si*ck,
    ss=si*sk).  Cross-checked against scipy.spatial.transform.Rotation (extrinsic xyz) at import."""
    from scipy.spatial.transform import Rotation
    m = types.ModuleType("transformations")

    def euler_matrix(ai, aj, ak, axes="sxyz"):
        assert axes == "sxyz"
        si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
        ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
        cc, cs = ci * ck, ci * sk
        sc, ss = si * ck, si * sk
        M = np.identity(4)
        M[0, 0] = cj * ck
        M[0, 1] = sj * sc - cs
        M[0, 2] = sj * cc + ss
        M[1, 0] = cj * sk
        M[1, 1] = sj * ss + cc
        M[1, 2] = sj * cs - sc
        M[2, 0] = -sj
        M[2, 1] = cj * si
        M[2, 2] = cj * ci
        return M
    for ang in ((0.3, -1.1, 2.5), (0, 0, math.pi / 2), (math.pi, 0.2, 0)):
        assert np.abs(euler_matrix(*ang)[:3, :3] - Rotation.from_euler("xyz", ang).as_matrix()).max() < 1e-15
    m.euler_matrix = euler_matrix
    m.__all__ = ["euler_matrix"]
    return m


def _cluster_poses(angle_diff, dist_diff, poses_in, symmetry_tfs):
    """mycpp.cluster_poses (pybind_api.cpp:24-68) in float32 numpy: greedy, first pose always kept, a pose joins the first
    cluster within dist_diff whose rotation is within angle_diff (degrees) of pose @ tf for some symmetry tf;
    rotationGeodesicDistance = acos(clamp((trace(R1 R2^T) - 1) / 2)) (mycpp/src/Utils.cpp:21-26)."""
    P = np.asarray(poses_in, dtype=np.float32)
    S = np.asarray(symmetry_tfs, dtype=np.float32)
    thres = np.float32(np.float32(angle_diff) / 180.0 * math.pi)
    out = [P[0]]
    for i in range(1, len(P)):
        cur, isnew = P[i], True
        for cl in out:
            if np.linalg.norm(cl[:3, 3] - cur[:3, 3]) >= dist_diff:
                continue
            for tf in S:
                R1 = (cur @ tf)[:3, :3]
                cs = np.float32((np.trace(R1 @ cl[:3, :3].T) - np.float32(1)) / np.float32(2.0))
                cs = max(min(cs, np.float32(1)), np.float32(-1))
                if np.arccos(cs) < thres:
                    isnew = False
                    break
            if not isnew:
                break
        if isnew:
            out.append(cur)
    return out


# ----------------------------------------------------------------------------------------------- torch 'cuda' -> cpu
def _cpu_device_patch():
    def fix(fn):
        def wrapped(*a, **k):
            if "device" in k and k["device"] is not None and "cuda" in str(k["device"]):
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped
    for name in ("as_tensor", "tensor", "eye", "ones", "zeros", "arange", "empty", "full", "linspace", "rand", "randn",
                 "from_numpy", "zeros_like", "ones_like"):
        setattr(torch, name, fix(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.set_default_tensor_type = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    _to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and "cuda" in x) else x for x in a)
        if "device" in k and "cuda" in str(k["device"]):
            k["device"] = "cpu"
        return _to(self, *a, **k)
    torch.Tensor.to = to


_loaded = None


def load_reference():
    """-> namespace with the reference modules imported on CPU (idempotent)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    _cpu_device_patch()
    mocks = ["imageio", "joblib", "open3d", "cv2", "ruamel", "ruamel.yaml", "torchvision",
             "h5py", "warp", "omegaconf", "pytorch3d.renderer", "pytorch3d.renderer.mesh", "pytorch3d.structures",
             "pytorch3d.renderer.mesh.rasterize_meshes", "pytorch3d.renderer.mesh.shader", "pytorch3d.renderer.mesh.textures",
             "bundlesdf", "bundlesdf.mycuda", "kaolin", "sklearn",
             "sklearn.metrics", "matplotlib", "matplotlib.pyplot", "pandas", "psutil", "yaml"]
    for m in mocks:
        if m not in sys.modules or m in ("yaml",):
            try:
                if m in ("pandas", "psutil", "yaml", "sklearn", "sklearn.metrics", "matplotlib", "matplotlib.pyplot", "joblib"):
                    importlib.import_module(m)
                    continue
            except Exception:
                pass
            sys.modules[m] = mock.MagicMock(name=m)
    tm = mock.MagicMock(name="trimesh")
    tm.creation.icosphere = _icosphere
    sys.modules["trimesh"] = tm
    sys.modules["transformations"] = _make_transformations()
    mc = types.ModuleType("mycpp.build.mycpp")
    mc.cluster_poses = _cluster_poses
    mcb = types.ModuleType("mycpp.build")
    mcb.mycpp = mc
    mcp = types.ModuleType("mycpp")
    mcp.build = mcb
    sys.modules["mycpp"], sys.modules["mycpp.build"], sys.modules["mycpp.build.mycpp"] = mcp, mcb, mc
    p3d = types.ModuleType("pytorch3d")
    p3dt = types.ModuleType("pytorch3d.transforms")
    for n in ("so3_log_map", "se3_exp_map", "se3_log_map", "matrix_to_axis_angle", "matrix_to_euler_angles", "euler_angles_to_matrix"):
        setattr(p3dt, n, mock.MagicMock(name=n))
    p3dt.so3_exp_map, p3dt.rotation_6d_to_matrix = _so3_exp_map, _rotation_6d_to_matrix
    p3d.transforms = p3dt
    sys.modules["pytorch3d"], sys.modules["pytorch3d.transforms"] = p3d, p3dt
    nv = types.ModuleType("nvdiffrast")
    nv.torch = _make_dr()
    sys.modules["nvdiffrast"], sys.modules["nvdiffrast.torch"] = nv, nv.torch
    k = _make_kornia()
    sys.modules["kornia"], sys.modules["kornia.geometry"], sys.modules["kornia.geometry.transform"] = k, k.geometry, k.geometry.transform
    if REF not in sys.path:
        sys.path.insert(0, REF)
    ns = types.SimpleNamespace()
    ns.Utils = importlib.import_module("Utils")
    ns.pose_dataset = importlib.import_module("learning.datasets.pose_dataset")
    ns.h5_dataset = importlib.import_module("learning.datasets.h5_dataset")
    ns.refine = importlib.import_module("learning.training.predict_pose_refine")
    ns.score = importlib.import_module("learning.training.predict_score")
    ns.refine_network = importlib.import_module("learning.models.refine_network")
    ns.score_network = importlib.import_module("learning.models.score_network")
    ns.estimater = importlib.import_module("estimater")
    _loaded = ns
    return ns
