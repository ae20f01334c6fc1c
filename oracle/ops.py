"""ctypes/numpy front-end of the C oracle (fp_oracle.c).  Test infrastructure only.

Every function mirrors one C-ABI entry point of the product library (include/fp_amd.h)
and cites the reference lines it restates in fp_oracle.c's header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_NORMALIZE_XYZ = 1
MODE_REFINE = 0
MODE_SCORE = 1
ROT_AXIS_ANGLE = 0
ROT_6D = 1


def build(force=False):
    so = os.path.join(_HERE, "libfp_oracle.so")
    src = os.path.join(_HERE, "fp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfp_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.fpo_num_threads.restype = C.c_int
        _LIB.fpo_cluster_poses.restype = C.c_int
    return _LIB


def num_threads():
    return int(lib().fpo_num_threads())


def set_num_threads(n):
    lib().fpo_set_num_threads(int(n))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def erode_depth(depth, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100.0):
    d = _f32(depth)
    out = np.empty_like(d)
    H, W = d.shape
    lib().fpo_erode_depth(_p(d), _p(out), H, W, int(radius), C.c_float(depth_diff_thres),
                          C.c_float(ratio_thres), C.c_float(zfar))
    return out


def bilateral_filter_depth(depth, radius=2, zfar=100.0, sigmaD=2.0, sigmaR=100000.0):
    d = _f32(depth)
    out = np.empty_like(d)
    H, W = d.shape
    lib().fpo_bilateral_depth(_p(d), _p(out), H, W, int(radius), C.c_float(zfar), C.c_float(sigmaD),
                              C.c_float(sigmaR))
    return out


def depth2xyzmap(depth, K, zfar=np.inf, f64_internal=True):
    d = _f32(depth)
    H, W = d.shape
    Kd = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    out = np.empty((H, W, 3), np.float32)
    lib().fpo_depth_to_xyz(_p(d), _p(Kd), C.c_float(zfar), int(bool(f64_internal)), _p(out), H, W)
    return out


def crop_windows(poses, K, mesh_diameter, crop_ratio, out_size=(160, 160)):
    """out_size = (width, height) as in compute_crop_window_tf_batch. -> tf_to_crops (N,3,3), bbox2d (N,4)"""
    P = _f32(poses).reshape(-1, 16)
    N = P.shape[0]
    Kd = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    tf = np.empty((N, 9), np.float32)
    bb = np.empty((N, 4), np.float32)
    lib().fpo_crop_windows(_p(P), _p(Kd), C.c_double(float(mesh_diameter)), C.c_double(float(crop_ratio)),
                           int(out_size[0]), int(out_size[1]), N, _p(tf), _p(bb))
    return tf.reshape(N, 3, 3), bb


def render_crops(mesh, poses, bbox2d, K, H, W, out_hw=(160, 160), mesh_diameter=1.0, xyz_thr=0.001,
                 normalize_xyz=True, w_ambient=0.8, w_diffuse=0.5, want=("A", "color", "depth", "xyz", "normal", "zbuf", "tri_id")):
    """mesh: dict with pos (V,3) f32, vnormals (V,3), faces (T,3) i32 and tex (Ht,Wt,3)+uv (V,2)[+uv_idx] or vertex_color (V,3)."""
    pos = _f32(mesh["pos"]); nrm = _f32(mesh["vnormals"])
    faces = np.ascontiguousarray(mesh["faces"], dtype=np.int32)
    V, T = pos.shape[0], faces.shape[0]
    tex = uv = uv_idx = vcol = None
    Ht = Wt = 0
    if mesh.get("tex") is not None:
        tex = _f32(mesh["tex"]).reshape(-1, mesh["tex"].shape[-2], 3)
        Ht, Wt = tex.shape[0], tex.shape[1]
        uv = _f32(mesh["uv"])
        if mesh.get("uv_idx") is not None:
            uv_idx = np.ascontiguousarray(mesh["uv_idx"], dtype=np.int32)
    else:
        vcol = _f32(mesh["vertex_color"])
    P = _f32(poses).reshape(-1, 16)
    N = P.shape[0]
    bb = None if bbox2d is None else _f32(bbox2d).reshape(N, 4)
    K9 = _f32(np.asarray(K, dtype=np.float64)).reshape(9)
    oh, ow = out_hw
    outs = {}
    def alloc(name, shape, dt):
        if name in want:
            outs[name] = np.empty(shape, dt)
            return outs[name]
        return None
    A = alloc("A", (N, 6, oh, ow), np.float32)
    color = alloc("color", (N, oh, ow, 3), np.float32)
    depth = alloc("depth", (N, oh, ow), np.float32)
    xyz = alloc("xyz", (N, oh, ow, 3), np.float32)
    normal = alloc("normal", (N, oh, ow, 3), np.float32)
    zbuf = alloc("zbuf", (N, oh, ow), np.uint32)
    tri = alloc("tri_id", (N, oh, ow), np.int32)
    lib().fpo_render_crops(_p(pos), _p(nrm), _p(faces), _p(uv), _p(uv_idx), _p(tex), Ht, Wt, _p(vcol), V, T,
                           _p(P), _p(bb), _p(K9), int(H), int(W), N, int(oh), int(ow),
                           C.c_float(w_ambient), C.c_float(w_diffuse), C.c_float(np.float32(mesh_diameter)),
                           C.c_float(xyz_thr), FLAG_NORMALIZE_XYZ if normalize_xyz else 0,
                           _p(A), _p(color), _p(depth), _p(xyz), _p(normal), _p(zbuf), _p(tri))
    return outs


def warp_crops(rgb, xyz_map, depth, tf_to_crops, K, poses, mesh_diameter, mode, normalize_xyz=True,
               out_hw=(160, 160)):
    rgbf = _f32(rgb)
    H, W = rgbf.shape[:2]
    xm = None if xyz_map is None else _f32(xyz_map)
    dp = None if depth is None else _f32(depth)
    tf = _f32(tf_to_crops).reshape(-1, 9)
    N = tf.shape[0]
    P = _f32(poses).reshape(N, 16)
    K9 = _f32(np.asarray(K, dtype=np.float64)).reshape(9)
    oh, ow = out_hw
    B = np.empty((N, 6, oh, ow), np.float32)
    lib().fpo_warp_crops(_p(rgbf), _p(xm), _p(dp), _p(tf), _p(K9), _p(P), C.c_float(np.float32(mesh_diameter)),
                         FLAG_NORMALIZE_XYZ if normalize_xyz else 0, int(mode), H, W, N, oh, ow, _p(B))
    return B


def pose_update(trans, rot, poses, rot_rep="axis_angle", normalize_xyz=True, trans_normalizer=(1, 1, 1),
                rot_normalizer=1.0, mesh_diameter=1.0):
    tr = _f32(trans); ro = _f32(rot)
    P = _f32(poses).reshape(-1, 16)
    N = P.shape[0]
    tn = _f32(np.broadcast_to(np.asarray(trans_normalizer, np.float32).reshape(-1), (3,)))
    out = np.empty_like(P)
    lib().fpo_pose_update(_p(tr), _p(ro), _p(P), ROT_AXIS_ANGLE if rot_rep == "axis_angle" else ROT_6D,
                          int(bool(normalize_xyz)), _p(tn), C.c_float(rot_normalizer),
                          C.c_float(np.float32(mesh_diameter)), N, _p(out))
    return out.reshape(N, 4, 4)


def deepim_trans_delta(trans, poses, K, tf_to_crops, input_w, normalize_xyz, mesh_diameter):
    """trans_rep='deepim' (predict_pose_refine.py:201-215): the network predicts the shift of the projected object
    centre in crop pixels (as a fraction of the crop width) and the ratio of the new depth to the current one;
    -> metric translation delta (N,3) f32.  All arithmetic in float32 like the reference's torch ops."""
    tr = _f32(trans)
    t = _f32(poses).reshape(-1, 4, 4)[:, :3, 3]
    K32 = np.asarray(K, dtype=np.float64).astype(np.float32)
    tf = _f32(tf_to_crops).reshape(-1, 3, 3)
    uv = (K32[None] @ t[:, :, None])[:, :, 0]
    uv = uv / uv[:, 2:3]
    uv_crop = (tf @ uv[:, :, None])[:, :2, 0]
    z_pred = tr[:, 2] * t[:, 2]
    uv_pred_crop = uv_crop + tr[:, :2] * np.float32(input_w)
    tfi = np.linalg.inv(tf).astype(np.float32)
    uv_pred = (tfi[:, :2, :2] @ uv_pred_crop[:, :, None])[:, :, 0] + tfi[:, :2, 2]
    cp = np.concatenate([uv_pred, np.ones((len(uv_pred), 1), np.float32)], axis=1)
    cp = (np.linalg.inv(K32).astype(np.float32)[None] @ cp[:, :, None])[:, :, 0] * z_pred[:, None]
    dt = (cp - t).astype(np.float32)
    if normalize_xyz:
        dt = dt * np.float32(float(mesh_diameter) / 2)
    return dt


def cluster_poses(angle_diff, dist_diff, poses, symmetry_tfs):
    P = _f32(poses).reshape(-1, 16)
    S = _f32(symmetry_tfs).reshape(-1, 16)
    keep = np.empty(P.shape[0], np.int32)
    nk = lib().fpo_cluster_poses(C.c_float(angle_diff), C.c_float(dist_diff), _p(P), P.shape[0], _p(S), S.shape[0], _p(keep))
    return keep[:nk].copy()
