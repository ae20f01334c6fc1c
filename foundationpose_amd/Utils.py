"""Host-side helpers with the names and argument meaning of the reference's Utils.py, re-implemented on top of
libfp_amd.so (device ops) and numpy (init-time setup).  Citations are to /root/reference/Utils.py."""
import logging
import random

import math

import numpy as np
import torch

from . import ops
from .mesh import SimpleMesh  # noqa: F401  (re-export)

glcam_in_cvcam = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]]).astype(float)  # Utils.py:68-71


def set_seed(random_seed):
    """Same effect as Utils.py:222-229: every RNG the pipeline can touch starts from `random_seed`, and the (unused here)
    cuDNN autotuner is pinned to deterministic choices so that a reference-side caller sees the flags it expects."""
    seed = int(random_seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True


# ------------------------------------------------------------------ mesh tensors (Utils.py:104-130)
def make_mesh_tensors(mesh, device="cuda", max_tex_size=None):
    """mesh: trimesh-like (``vertices``, ``faces``, ``vertex_normals``, ``visual``).  Returns the reference's dict of
    device tensors plus '_handle' (the fp_mesh used by the HIP rasteriser)."""
    visual = getattr(mesh, "visual", None)
    material = getattr(visual, "material", None)
    image = getattr(material, "image", None) if material is not None else None
    t = {}
    if image is not None and getattr(visual, "uv", None) is not None:
        img = image.convert("RGB") if hasattr(image, "convert") else image
        img = np.asarray(img)[..., :3]
        if max_tex_size is not None and max(img.shape[:2]) > max_tex_size:
            step = int(np.ceil(max(img.shape[:2]) / max_tex_size))
            img = img[::step, ::step]  # decimation; the reference uses cv2.resize (Utils.py:110-114, setup-time only)
        t["tex"] = torch.as_tensor(np.ascontiguousarray(img), device=device, dtype=torch.float)[None] / 255.0
        t["uv_idx"] = torch.as_tensor(np.asarray(mesh.faces), device=device, dtype=torch.int)
        uv = torch.as_tensor(np.asarray(visual.uv), device=device, dtype=torch.float)
        uv[:, 1] = 1 - uv[:, 1]
        t["uv"] = uv
    else:
        vc = getattr(visual, "vertex_colors", None) if visual is not None else None
        if vc is None:
            logging.info("WARN: mesh doesn't have vertex_colors, assigning a pure color")
            vc = np.tile(np.array([128, 128, 128]).reshape(1, 3), (len(mesh.vertices), 1))
            if visual is not None:
                try:
                    visual.vertex_colors = vc  # the reference mutates the mesh too (Utils.py:120-122)
                except Exception:
                    pass
        t["vertex_color"] = torch.as_tensor(np.asarray(vc)[..., :3], device=device, dtype=torch.float) / 255.0
    t["pos"] = torch.tensor(np.asarray(mesh.vertices), device=device, dtype=torch.float)
    t["faces"] = torch.tensor(np.asarray(mesh.faces), device=device, dtype=torch.int)
    t["vnormals"] = torch.tensor(np.asarray(mesh.vertex_normals), device=device, dtype=torch.float)
    t["_handle"] = mesh_handle_from_tensors(t)
    return t


def mesh_handle_from_tensors(t):
    return ops.MeshHandle(t["pos"].contiguous(), t["vnormals"].contiguous(), t["faces"].contiguous(),
                          uv=t["uv"].contiguous() if "uv" in t else None,
                          uv_idx=t["uv_idx"].contiguous() if "uv_idx" in t else None,
                          tex=t["tex"].contiguous() if "tex" in t else None,
                          vertex_color=t["vertex_color"].contiguous() if "vertex_color" in t else None)


def get_mesh_handle(mesh_tensors):
    h = mesh_tensors.get("_handle")
    if h is None or h.device != mesh_tensors["pos"].device or h.pos.data_ptr() != mesh_tensors["pos"].data_ptr():
        h = mesh_handle_from_tensors(mesh_tensors)
        mesh_tensors["_handle"] = h
    return h


# ------------------------------------------------------------------ render (Utils.py:133-219)
def nvdiffrast_render(K=None, H=None, W=None, ob_in_cams=None, glctx=None, context="cuda", get_normal=False,
                      mesh_tensors=None, mesh=None, projection_mat=None, bbox2d=None, output_size=None,
                      use_light=False, light_color=None, light_dir=np.array([0, 0, 1]), light_pos=np.array([0, 0, 0]),
                      w_ambient=0.8, w_diffuse=0.5, extra={}):
    """Same contract as the reference wrapper: returns (color (N,h,w,3) in [0,1], depth (N,h,w), normal_map or None)
    and fills extra['xyz_map'].  Rasterisation, interpolation, texturing, shading and the Y flips are one HIP kernel."""
    if context not in ("cuda", "gl"):
        raise NotImplementedError
    if projection_mat is not None:
        raise NotImplementedError("custom projection_mat is not supported; the projection is derived from K,H,W")
    if light_color is not None or light_dir is None:
        raise NotImplementedError("only the default directional light of the hot path is implemented")
    if mesh_tensors is None:
        mesh_tensors = make_mesh_tensors(mesh)
    handle = get_mesh_handle(mesh_tensors)
    poses = torch.as_tensor(ob_in_cams, device=handle.device, dtype=torch.float).reshape(-1, 4, 4).contiguous()
    if output_size is None:
        output_size = (H, W)
    bb = None
    if bbox2d is not None:
        bb = torch.as_tensor(bbox2d, device=handle.device, dtype=torch.float).reshape(-1, 4)
        if bb.shape[0] != poses.shape[0]:
            bb = bb.expand(poses.shape[0], 4)
        bb = bb.contiguous()
    elif (int(output_size[0]), int(output_size[1])) != (int(H), int(W)):
        raise NotImplementedError("output_size != (H,W) needs bbox2d")
    if use_light:
        get_normal = True
    want = ["color", "depth", "xyz"] + (["normal"] if get_normal else [])
    # without lighting the reference returns the unshaded base colour: ambient 1, diffuse 0
    wa, wd = (w_ambient, w_diffuse) if use_light else (1.0, 0.0)
    outs = []
    for b in range(0, poses.shape[0], 4096):
        outs.append(ops.render_crops(handle, poses[b:b + 4096], None if bb is None else bb[b:b + 4096], K, H, W,
                                     out_hw=(int(output_size[0]), int(output_size[1])), want=want, w_ambient=wa,
                                     w_diffuse=wd))
    cat = (lambda k: outs[0][k] if len(outs) == 1 else torch.cat([o[k] for o in outs], dim=0))
    extra["xyz_map"] = cat("xyz")
    return cat("color"), cat("depth"), (cat("normal") if get_normal else None)


# ------------------------------------------------------------------ depth filters / back-projection
def _to_dev_f32(x, device="cuda"):
    return torch.as_tensor(x, dtype=torch.float, device=device).contiguous()


def erode_depth(depth, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100, device="cuda"):
    """Utils.py:387-395: numpy in => numpy out, tensor in => tensor out."""
    out = ops.erode_depth(_to_dev_f32(depth, device), radius, depth_diff_thres, ratio_thres, float(zfar))
    return out.data.cpu().numpy() if isinstance(depth, np.ndarray) else out


def bilateral_filter_depth(depth, radius=2, zfar=100, sigmaD=2, sigmaR=100000, device="cuda"):
    """Utils.py:345-356"""
    out = ops.bilateral_filter_depth(_to_dev_f32(depth, device), radius, float(zfar), float(sigmaD), float(sigmaR))
    return out.data.cpu().numpy() if isinstance(depth, np.ndarray) else out


def depth2xyzmap(depth, K, uvs=None):
    """Utils.py:399-417 (float64 intrinsics).  numpy in => numpy out, tensor in => tensor out."""
    if uvs is not None:
        raise NotImplementedError("sparse uvs are not on the hot path")
    out = ops.depth_to_xyz(_to_dev_f32(depth), K, zfar=float("inf"), f64_internal=True)
    return out.data.cpu().numpy() if isinstance(depth, np.ndarray) else out


def depth2xyzmap_batch(depths, Ks, zfar):
    """Utils.py:420-438: depths (B,H,W) tensor, Ks (B,3,3) tensor -> (B,H,W,3)."""
    depths = _to_dev_f32(depths)
    Ks = torch.as_tensor(Ks).detach().cpu().numpy()
    return torch.stack([ops.depth_to_xyz(depths[b], Ks[b], zfar=float(zfar), f64_internal=False)
                        for b in range(depths.shape[0])], dim=0)


# ------------------------------------------------------------------ small transforms
def warp_perspective_nearest(src, M, dsize):
    """kornia.geometry.transform.warp_perspective(src, M, dsize, mode='nearest', align_corners=False) for the ONE call
    site family that is not on the hot path: the normal-map warps of `use_normal=True` (predict_pose_refine.py:75-76), whose
    results the reference stores in BatchPoseData and never feeds to a network.  kornia 0.7.2 semantics [3P, SURVEY App.
    B.2]: the homography is conjugated with the [0, size-1] -> [-1, 1] pixel normalisations of source and destination,
    inverted, applied to a linspace(-1, 1) grid, and handed to F.grid_sample (zeros padding).  Plain torch ops on the
    tensors' device.  src (B,C,H,W), M (B,3,3) source-pixel -> destination-pixel."""
    import torch.nn.functional as F
    B, _, H, W = src.shape
    h, w = int(dsize[0]), int(dsize[1])

    def norm_px(hh, ww):
        return torch.tensor([[2.0 / (ww - 1), 0.0, -1.0], [0.0, 2.0 / (hh - 1), -1.0], [0.0, 0.0, 1.0]], dtype=M.dtype, device=M.device)
    dst_from_src = norm_px(h, w)[None] @ (M @ torch.linalg.inv(norm_px(H, W))[None])
    src_from_dst = torch.linalg.inv(dst_from_src)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h, dtype=M.dtype, device=M.device),
                            torch.linspace(-1, 1, w, dtype=M.dtype, device=M.device), indexing="ij")
    grid = torch.stack([xs, ys, torch.ones_like(xs)], -1).reshape(1, -1, 3) @ src_from_dst.transpose(1, 2)
    grid = (grid[..., :2] / grid[..., 2:3]).reshape(B, h, w, 2)
    return F.grid_sample(src, grid.to(src.dtype), mode="nearest", padding_mode="zeros", align_corners=False)


def to_homo_torch(pts):
    """(..., d) -> (..., d+1) with a trailing 1 (Utils.py:520-526)"""
    return torch.nn.functional.pad(pts.to(torch.float), (0, 1), value=1.0)


def _per_point(tf, n_points):
    """Broadcasting rule of the reference's transform helpers: a batch of transforms whose batch dimension is not the
    point count gets a singleton point axis, i.e. every transform is applied to all points.  (A batch as long as the
    point list pairs transform i with point i: SURVEY App. D.5 -- reproduced, callers rely on it.)"""
    return tf.unsqueeze(-3) if tf.dim() >= 3 and tf.shape[-3] != n_points else tf


def transform_pts(pts, tf):
    """Rigid / affine map of points, x -> R x + t with [R | t] the top rows of tf (semantics of Utils.py:529-536)"""
    tf = _per_point(tf, pts.shape[-2])
    d = pts.shape[-1]
    return torch.einsum("...ij,...j->...i", tf[..., :d, :d], pts) + tf[..., :d, d]


def transform_dirs(dirs, tf):
    """Directions only see the rotation block (semantics of Utils.py:539-546)"""
    return torch.einsum("...ij,...j->...i", _per_point(tf, dirs.shape[-2])[..., :3, :3], dirs)


def egocentric_delta_pose_to_pose(A_in_cam, trans_delta, rot_mat_delta):
    """Utils.py:848-855 semantics: the rotation delta acts in the camera frame about the object origin
    (R' = dR R), the translation delta is added (t' = t + dt); fp_pose_update is the fused device version."""
    out = torch.zeros((A_in_cam.shape[0], 4, 4), dtype=torch.float, device=A_in_cam.device)
    out[:, :3, :3] = torch.bmm(rot_mat_delta.to(torch.float), A_in_cam[:, :3, :3].to(torch.float))
    out[:, :3, 3] = A_in_cam[:, :3, 3] + trans_delta
    out[:, 3, 3] = 1.0
    return out


def compute_crop_window_tf_batch(pts=None, H=None, W=None, poses=None, K=None, crop_ratio=1.2, out_size=None, rgb=None,
                                 uvs=None, method="min_box", mesh_diameter=None):
    """Utils.py:577-626 (only method='box_3d' exists there).  Returns tf_to_crops (B,3,3) on the device."""
    if method != "box_3d":
        raise RuntimeError
    poses = torch.as_tensor(poses, dtype=torch.float, device="cuda").reshape(-1, 4, 4).contiguous()
    tf, _ = ops.crop_windows(poses, K, mesh_diameter, crop_ratio, out_size)
    return tf


def projection_matrix_from_intrinsics(K, height, width, znear, zfar, window_coords="y_down"):
    """Utils.py:752-802 (kept for API completeness; the HIP rasteriser projects with K directly)."""
    K = np.asarray(K, dtype=float)
    depth = float(zfar - znear)
    q = -(zfar + znear) / depth
    qn = -2 * (zfar * znear) / depth
    if window_coords == "y_up":
        r1 = [0, -2 * K[1, 1] / height, (-2 * K[1, 2] + height) / height, 0]
    elif window_coords == "y_down":
        r1 = [0, 2 * K[1, 1] / height, (2 * K[1, 2] - height) / height, 0]
    else:
        raise NotImplementedError
    return np.array([[2 * K[0, 0] / width, -2 * K[0, 1] / width, (-2 * K[0, 2] + width) / width, 0], r1,
                     [0, 0, q, qn], [0, 0, -1, 0]])


# ------------------------------------------------------------------ hypothesis-generation setup (CPU, init time)
def euler_matrix(ai, aj, ak, axes="sxyz"):
    """Static-xyz Euler angles -> 4x4, entry by entry as transformations.euler_matrix (Gohlke's transformations.py [3P],
    imported by the reference's Utils.py:34) computes it for its default axes: R = Rz(ak) Ry(aj) Rx(ai) written out in
    products of the six sines / cosines (so that e.g. a 90 degree step carries the same 6e-17 residues)."""
    if axes != "sxyz":
        raise NotImplementedError
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    M = np.eye(4)
    M[0, 0], M[0, 1], M[0, 2] = cj * ck, sj * sc - cs, sj * cc + ss
    M[1, 0], M[1, 1], M[1, 2] = cj * sk, sj * ss + cc, sj * cs - sc
    M[2, 0], M[2, 1], M[2, 2] = -sj, cj * si, cj * ci
    return M


def icosphere_vertices(subdivisions=1, radius=1.0):
    """Vertices of trimesh.creation.icosphere(subdivisions, radius) [3P: trimesh 4.2.2 is not installed here; restated from
    its published algorithm -- creation.icosahedron / remesh.subdivide / the spherical re-projection -- so that the ORDER
    of the views, hence of the 252 hypotheses (estimater.py:106-124) and of what the greedy cluster_poses keeps for a
    symmetric object, is the reference's]:
      * the 12 icosahedron vertices (+-1, +-t, 0) cyclic, divided by sqrt(2 + t);
      * per level: one midpoint per unique edge, appended in the order of trimesh's row hash of the sorted edge (lo, hi) --
        lo | hi << 32, i.e. by hi, then lo; then every vertex is moved to the sphere by v += v/|v| * (radius - |v|)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([-1, t, 0, 1, t, 0, -1, -t, 0, 1, -t, 0, 0, -1, t, 0, 1, t, 0, -1, -t, 0, 1, -t, t, 0, -1, t, 0, 1, -t, 0, -1,
                  -t, 0, 1], dtype=np.float64).reshape(-1, 3) / np.sqrt(2.0 + t)
    f = np.array([0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8, 3, 9, 4, 3, 4,
                  2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1], dtype=np.int64).reshape(-1, 3)
    for _ in range(subdivisions):
        edges = np.sort(f[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2), axis=1)
        _, first, inverse = np.unique(edges[:, 0] | (edges[:, 1] << 32), return_index=True, return_inverse=True)
        mid = v[edges[first]].mean(axis=1)
        m = inverse.reshape(-1, 3) + len(v)
        f = np.column_stack([f[:, 0], m[:, 0], m[:, 2], m[:, 0], f[:, 1], m[:, 1], m[:, 2], m[:, 1], f[:, 2], m[:, 0], m[:, 1],
                             m[:, 2]]).reshape(-1, 3)
        v = np.vstack((v, mid))
        scalar = np.sqrt(np.dot(v ** 2, [1, 1, 1]))
        v = v + (v / scalar.reshape(-1, 1)) * (radius - scalar).reshape(-1, 1)
    return v


def sample_views_icosphere(n_views, subdivisions=None, radius=1):
    """Camera-in-object poses on an icosphere, each looking at the origin (semantics of Utils.py:483-507): camera z
    points from the vertex to the centre, x = world-up (0,0,1) x z (or (1,0,0) at the poles), y = z x x."""
    if subdivisions is None:
        subdivisions = 1
        while icosphere_vertices(subdivisions, radius).shape[0] < n_views:
            subdivisions += 1
    eye = icosphere_vertices(subdivisions, radius)
    fwd = -eye / np.linalg.norm(eye, axis=1, keepdims=True)
    right = np.cross(np.array([[0.0, 0.0, 1.0]]), fwd)
    right[~right.any(axis=1)] = (1.0, 0.0, 0.0)        # looking along the up axis: any perpendicular will do, the reference picks +x
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    down /= np.linalg.norm(down, axis=1, keepdims=True)
    poses = np.zeros((len(eye), 4, 4))
    poses[:, :3, :] = np.stack([right, down, fwd, eye], axis=2)
    poses[:, 3, 3] = 1.0
    return poses


def compute_mesh_diameter(model_pts=None, mesh=None, n_sample=1000):
    """Utils.py:559-574: max pairwise distance of (a random subset of) the points; returns np.float64."""
    if mesh is not None:
        model_pts = np.asarray(mesh.vertices)
    pts = np.asarray(model_pts, dtype=np.float64)
    if n_sample is not None:
        ids = np.random.choice(len(pts), size=min(n_sample, len(pts)), replace=False)
        pts = pts[ids]
    best = 0.0
    for i in range(0, len(pts), 1024):
        d = np.linalg.norm(pts[None] - pts[i:i + 1024, None], axis=-1)
        best = max(best, d.max())
    return np.float64(best)


def symmetry_tfs_from_info(info, rot_angle_discrete=5):
    """BOP models_info entry -> (S,4,4) symmetry transforms in metres, identity first (semantics of Utils.py:806-834):
    the listed discrete symmetries (translations are in mm), then, for the FIRST continuous symmetry, rotations about the
    coordinate axis its `axis` vector is positive in (x before y before z), every `rot_angle_discrete` degrees starting at
    0, each carrying the symmetry's `offset` as translation."""
    out = [np.eye(4)]
    discrete = info.get("symmetries_discrete")
    if discrete is not None:
        for m in np.asarray(discrete, dtype=float).reshape(-1, 4, 4):
            m = m.copy()
            m[:3, 3] *= 0.001
            out.append(m)
    continuous = info.get("symmetries_continuous")
    if continuous:
        sym = continuous[0]
        axis = np.asarray(sym["axis"], dtype=float).reshape(3)
        positive = [i for i in range(3) if axis[i] > 0]
        angles = np.arange(0, 360, rot_angle_discrete) / 180.0 * np.pi        # Utils.py:820-824
        for ang in (angles if positive else [0.0]):
            euler = [0.0, 0.0, 0.0]
            if positive:
                euler[positive[0]] = ang
            m = euler_matrix(*euler)
            m[:3, 3] = sym["offset"]
            out.append(m)
    return np.array(out)


def cluster_poses(angle_diff, dist_diff, poses_in, symmetry_tfs):
    """mycpp.cluster_poses look-alike (mycpp/src/app/pybind_api.cpp:24-68): returns the kept poses."""
    poses_in = np.asarray(poses_in)
    keep = ops.cluster_poses(angle_diff, dist_diff, poses_in, symmetry_tfs)
    return poses_in[keep]
