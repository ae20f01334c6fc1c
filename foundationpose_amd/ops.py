"""torch-tensor front-end of the C ABI (device pointers + current HIP stream).  Plumbing only."""
import ctypes as C

import numpy as np
import torch

from . import _lib

FLAG_NORMALIZE_XYZ = 1
FLAG_OUT_F16 = 2
MODE_REFINE = 0
MODE_SCORE = 1
ROT_AXIS_ANGLE = 0
ROT_6D = 1
TRANS_TRACKNET, TRANS_DEEPIM, TRANS_RAW = 0, 1, 2


def _stream(t=None):
    """HIP stream the launch goes to: PyTorch's current stream of the tensor's device.  A tensor on another device than
    the thread's current one is refused (launches go to the current device): use torch.cuda.set_device / torch.cuda.device."""
    if t is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dev = t.device
    if dev.index is not None and dev.index != torch.cuda.current_device():
        raise _lib.FpAmdError(f"tensor on {dev} but the current device is cuda:{torch.cuda.current_device()}: wrap the call in "
                              f"`with torch.cuda.device({dev.index}):` (kernels launch on the current device)")
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _dev(t, dtype, name):
    if t is None:
        return None
    if not (torch.is_tensor(t) and t.is_cuda):
        raise _lib.FpAmdError(f"{name}: expected a CUDA(HIP) tensor; there is no CPU path in foundationpose_amd")
    if t.dtype != dtype:
        raise _lib.FpAmdError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.FpAmdError(f"{name}: tensor must be contiguous")
    return t


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _hostK64(K):
    return np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(9))


def _hostK32(K):
    if torch.is_tensor(K):
        K = K.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(9).astype(np.float32))


class MeshHandle:
    """Device mesh tensors + the fp_mesh handle (Utils.py:104-130 make_mesh_tensors)."""

    def __init__(self, pos, vnormals, faces, uv=None, uv_idx=None, tex=None, vertex_color=None):
        self.pos = _dev(pos, torch.float32, "pos")
        self.vnormals = _dev(vnormals, torch.float32, "vnormals")
        self.faces = _dev(faces, torch.int32, "faces")
        self.uv = _dev(uv, torch.float32, "uv")
        self.uv_idx = _dev(uv_idx, torch.int32, "uv_idx")
        self.tex = _dev(tex, torch.float32, "tex")
        self.vertex_color = _dev(vertex_color, torch.float32, "vertex_color")
        self.V, self.T = int(pos.shape[0]), int(faces.shape[0])
        Ht = Wt = 0
        if self.tex is not None:
            Ht, Wt = int(self.tex.shape[-3]), int(self.tex.shape[-2])
        h = C.c_void_p()
        st = _lib.lib().fp_mesh_create(_ptr(self.pos), _ptr(self.vnormals), _ptr(self.faces), _ptr(self.uv),
                                       _ptr(self.uv_idx), _ptr(self.tex), _ptr(self.vertex_color), self.V, self.T,
                                       Ht, Wt, C.byref(h))
        _lib.check(st, "fp_mesh_create")
        self.handle = h
        self.device = self.pos.device

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().fp_mesh_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def erode_depth(depth, radius=2, depth_diff_thres=0.001, ratio_thres=0.8, zfar=100.0):
    d = _dev(depth, torch.float32, "depth")
    out = torch.empty_like(d)
    H, W = d.shape
    _lib.check(_lib.lib().fp_depth_erode(_ptr(d), _ptr(out), H, W, int(radius), depth_diff_thres, ratio_thres, zfar,
                                         _stream(d)), "fp_depth_erode")
    return out


def bilateral_filter_depth(depth, radius=2, zfar=100.0, sigmaD=2.0, sigmaR=100000.0):
    d = _dev(depth, torch.float32, "depth")
    out = torch.empty_like(d)
    H, W = d.shape
    _lib.check(_lib.lib().fp_depth_bilateral(_ptr(d), _ptr(out), H, W, int(radius), zfar, sigmaD, sigmaR, _stream(d)),
               "fp_depth_bilateral")
    return out


def depth_to_xyz(depth, K, zfar=float("inf"), f64_internal=False):
    d = _dev(depth, torch.float32, "depth")
    H, W = d.shape
    out = torch.empty((H, W, 3), dtype=torch.float32, device=d.device)
    Kd = _hostK64(K)
    _lib.check(_lib.lib().fp_depth_to_xyz(_ptr(d), Kd.ctypes.data_as(C.c_void_p), float(zfar), int(bool(f64_internal)),
                                          _ptr(out), H, W, _stream(d)), "fp_depth_to_xyz")
    return out


def crop_windows(poses, K, mesh_diameter, crop_ratio, out_size=(160, 160)):
    """-> tf_to_crops (N,3,3) f32, bbox2d (N,4) f32.  out_size = (width, height)."""
    P = _dev(poses, torch.float32, "poses")
    N = int(P.shape[0])
    tf = torch.empty((N, 3, 3), dtype=torch.float32, device=P.device)
    bb = torch.empty((N, 4), dtype=torch.float32, device=P.device)
    Kd = _hostK64(K)
    _lib.check(_lib.lib().fp_crop_windows(_ptr(P), Kd.ctypes.data_as(C.c_void_p), float(mesh_diameter),
                                          float(crop_ratio), int(out_size[0]), int(out_size[1]), N, _ptr(tf), _ptr(bb),
                                          _stream(P)), "fp_crop_windows")
    return tf, bb


_WS = {}


def workspace_bytes(N, V, T, oh=160, ow=160):
    return int(_lib.lib().fp_workspace_bytes(int(N), int(V), int(T), int(oh), int(ow)))


def _workspace(nbytes, device):
    """library-side default scratch: one per (device, stream) -- launches on different streams may overlap.

    Inside a stream capture the stream-keyed scratch is NEVER handed out (round 5, the advisor's finding: torch.cuda.graph uses one
    shared capture stream, so two graphs captured one after the other baked in the SAME scratch address; replayed on different
    streams -- the PartGraphs pattern -- they raced on it and tri_id / zbuf came out silently wrong).  A capture without a caller-owned
    `workspace` gets a fresh allocation made inside the capture: it comes from the capturing graph's private pool, lives exactly as
    long as that graph, and no other graph or eager launch can hold its address."""
    if nbytes == 0:
        return None
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def render_crops(mesh, poses, bbox2d, K, H, W, out_hw=(160, 160), mesh_diameter=1.0, xyz_thr=0.001,
                 normalize_xyz=True, out_f16=False, w_ambient=0.8, w_diffuse=0.5,
                 want=("A",), A_out=None, workspace=None):
    """Fused render of N hypotheses (see fp_render_crops).  Returns dict of requested outputs.  workspace: caller-owned
    uint8 scratch of at least workspace_bytes(...) bytes (a captured hipGraph must own its scratch); default: a
    per-device scratch that grows on demand."""
    P = _dev(poses, torch.float32, "poses")
    N = int(P.shape[0])
    bb = _dev(bbox2d, torch.float32, "bbox2d")
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = P.device
    outs = {}

    def alloc(name, shape, dt):
        if name in want:
            outs[name] = torch.empty(shape, dtype=dt, device=dev)
            return outs[name]
        return None

    if A_out is not None:
        A = A_out
        outs["A"] = A
    else:
        A = alloc("A", (N, 6, oh, ow), torch.float16 if out_f16 else torch.float32)
    color = alloc("color", (N, oh, ow, 3), torch.float32)
    depth = alloc("depth", (N, oh, ow), torch.float32)
    xyz = alloc("xyz", (N, oh, ow, 3), torch.float32)
    normal = alloc("normal", (N, oh, ow, 3), torch.float32)
    zbuf = alloc("zbuf", (N, oh, ow), torch.int32)  # u32 payload, viewed as int32 by torch
    tri = alloc("tri_id", (N, oh, ow), torch.int32)
    L = _lib.lib()
    need = L.fp_workspace_bytes(N, mesh.V, mesh.T, oh, ow)
    if workspace is not None:
        ws = _dev(workspace, torch.uint8, "workspace")
        if ws.numel() < need:
            raise _lib.FpAmdError(f"render_crops: workspace has {ws.numel()} bytes, {need} needed")
    else:
        ws = _workspace(need, dev)
    K9 = _hostK32(K)
    flags = (FLAG_NORMALIZE_XYZ if normalize_xyz else 0) | (FLAG_OUT_F16 if (A is not None and A.dtype == torch.float16) else 0)
    st = L.fp_render_crops(mesh.handle, _ptr(P), _ptr(bb), K9.ctypes.data_as(C.c_void_p), int(H), int(W), N, oh, ow,
                           w_ambient, w_diffuse, float(np.float32(mesh_diameter)), xyz_thr, flags, _ptr(A), _ptr(color),
                           _ptr(depth), _ptr(xyz), _ptr(normal), _ptr(zbuf), _ptr(tri), _ptr(ws),
                           0 if ws is None else ws.numel(), _stream(P))
    _lib.check(st, "fp_render_crops")
    return outs


def warp_crops(rgb, xyz_map, depth, tf_to_crops, K, poses, mesh_diameter, mode, normalize_xyz=True, out_f16=False,
               out_hw=(160, 160), B_out=None):
    rgbf = _dev(rgb, torch.float32, "rgb")
    H, W = int(rgbf.shape[0]), int(rgbf.shape[1])
    xm = _dev(xyz_map, torch.float32, "xyz_map")
    dp = _dev(depth, torch.float32, "depth")
    tf = _dev(tf_to_crops, torch.float32, "tf_to_crops")
    P = _dev(poses, torch.float32, "poses")
    N = int(P.shape[0])
    oh, ow = int(out_hw[0]), int(out_hw[1])
    B = B_out if B_out is not None else torch.empty((N, 6, oh, ow), dtype=torch.float16 if out_f16 else torch.float32,
                                                    device=P.device)
    flags = (FLAG_NORMALIZE_XYZ if normalize_xyz else 0) | (FLAG_OUT_F16 if B.dtype == torch.float16 else 0)
    K9 = _hostK32(K)
    st = _lib.lib().fp_warp_crops(_ptr(rgbf), _ptr(xm), _ptr(dp), _ptr(tf), K9.ctypes.data_as(C.c_void_p), _ptr(P),
                                  float(np.float32(mesh_diameter)), flags, int(mode), H, W, N, oh, ow, _ptr(B), _stream(P))
    _lib.check(st, "fp_warp_crops")
    return B


def pose_update(trans, rot, poses, rot_rep="axis_angle", normalize_xyz=True, trans_normalizer=(1.0, 1.0, 1.0),
                rot_normalizer=1.0, mesh_diameter=1.0, out=None, trans_delta_out=None, rot_delta_out=None, trans_rep="tracknet",
                K=None, tf_to_crops=None, input_w=0):
    """fp_pose_update.  trans_rep='deepim' needs K, tf_to_crops (N,3,3) and the crop width (predict_pose_refine.py:201-215);
    any trans_rep other than 'tracknet' / 'deepim' is the reference's plain `else` branch (:217-218): the raw output"""
    tr = _dev(trans, torch.float32, "trans")
    ro = _dev(rot, torch.float32, "rot")
    P = _dev(poses, torch.float32, "poses")
    N = int(P.shape[0])
    if rot_rep == "axis_angle":
        rr = ROT_AXIS_ANGLE
    elif rot_rep == "6d":
        rr = ROT_6D
    else:
        raise RuntimeError(f"unknown rot_rep {rot_rep}")
    tn = np.ascontiguousarray(np.broadcast_to(np.asarray(trans_normalizer, dtype=np.float32).reshape(-1), (3,)))
    O = out if out is not None else torch.empty_like(P)
    deepim = trans_rep == "deepim"
    K9 = _hostK32(K) if deepim else None
    tf = _dev(tf_to_crops, torch.float32, "tf_to_crops") if deepim else None
    st = _lib.lib().fp_pose_update(_ptr(tr), _ptr(ro), _ptr(P), rr, int(bool(normalize_xyz)),
                                   tn.ctypes.data_as(C.c_void_p), float(rot_normalizer), float(np.float32(mesh_diameter)),
                                   N, _ptr(O), _ptr(_dev(trans_delta_out, torch.float32, "trans_delta_out")),
                                   _ptr(_dev(rot_delta_out, torch.float32, "rot_delta_out")),
                                   TRANS_DEEPIM if deepim else (TRANS_TRACKNET if trans_rep == "tracknet" else TRANS_RAW),
                                   K9.ctypes.data_as(C.c_void_p) if deepim else None, _ptr(tf), float(input_w), _stream(P))
    _lib.check(st, "fp_pose_update")
    return O


def conv7x7s2_bn_relu(x, w_flat, bias, scale, shift, out, pad):
    """x (B,6,H,W) f16 NCHW -> interior of `out` (B, H/2 + 2 pad, W/2 + 2 pad, 64) f16 NHWC (border untouched), following
    the autocast op sequence conv -> fp16, + bias -> fp16, eval BatchNorm -> fp16, ReLU (fp_conv7x7s2_bn_relu_fwd).
    bias (fp16-representable values), scale, shift: (64) f32 or None."""
    x = _dev(x, torch.float16, "x")
    w = _dev(w_flat, torch.float16, "w")
    y = _dev(out, torch.float16, "out")
    Bn, Cin, H, W = x.shape
    if Cin != 6:
        raise _lib.FpAmdError("conv7x7s2_bn_relu: C_in must be 6")
    if tuple(y.shape) != (Bn, H // 2 + 2 * pad, W // 2 + 2 * pad, 64):
        raise _lib.FpAmdError(f"conv7x7s2_bn_relu: out has shape {tuple(y.shape)}")
    st = _lib.lib().fp_conv7x7s2_bn_relu_fwd(_ptr(x), _ptr(w), _ptr(_dev(bias, torch.float32, "bias")),
                                             _ptr(_dev(scale, torch.float32, "scale")), _ptr(_dev(shift, torch.float32, "shift")),
                                             _ptr(y), int(Bn), int(H), int(W), int(pad), _stream(x))
    _lib.check(st, "fp_conv7x7s2_bn_relu_fwd")
    return y


class IgemmGeom(C.Structure):
    """fp_igemm_geom (include/fp_amd.h): addressing of one NHWC fp16 operand of fp_igemm_f16_fwd"""
    _fields_ = [(n, C.c_int) for n in ("pixels_per_image", "width", "padded_h", "padded_w", "stride", "offset", "cstride",
                                       "coff", "bsplit", "cgroup")]

    @staticmethod
    def matrix(ld):
        return IgemmGeom(1, 1, 1, 1, 1, 0, int(ld), 0, 0, 0)

    @staticmethod
    def image(Ho, Wo, pad, C_, stride=1, offset=None, coff=0, bsplit=0, cgroup=0):
        """rows = output pixels (Ho x Wo per image) addressed in a buffer (B, Ho*stride + 2*pad, Wo*stride + 2*pad, C_)
        (for stride 1 that is the output / residual buffer itself; for the conv INPUT pass offset=0 so that the
        geometry addresses tap (0,0))"""
        return IgemmGeom(Ho * Wo, Wo, Ho * stride + 2 * pad, Wo * stride + 2 * pad, stride, pad if offset is None else offset,
                         C_, coff, bsplit, cgroup)


IGEMM_RELU = 1
IGEMM_ROUND_ACC = 2
IGEMM_HAS_W_TILES = 4


class IgemmEpilogue(C.Structure):
    """fp_igemm_epilogue (include/fp_amd.h)"""
    _fields_ = [("bias", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("residual", C.c_void_p),
                ("r_geom", C.POINTER(IgemmGeom)), ("flags", C.c_int), ("pe", C.c_void_p), ("pe_period", C.c_int), ("y_pe", C.c_void_p),
                ("w_tiles", C.c_void_p)]


def pack_conv3x3_tiles(w, N, Cin):
    """(N, 9 * Cin) fp16 conv weight, k ordered (ky, kx, ci) -> the tile-packed copy fp_igemm_f16_fwd's shifted-window kernel reads as
    contiguous 8 KiB runs (fp_pack_conv3x3_tiles_f16); built once per weight by a plan"""
    w = _dev(w, torch.float16, "w")
    out = torch.empty_like(w)
    _lib.check(_lib.lib().fp_pack_conv3x3_tiles_f16(_ptr(w), _ptr(out), int(N), int(Cin), _stream(w)), "fp_pack_conv3x3_tiles_f16")
    return out


def igemm_f16(x, x_geom, w, bias, y, y_geom, M, N, Cin, taps, relu=False, residual=None, r_geom=None, bn_scale=None,
              bn_shift=None, conv_rounding=False, pe=None, y_pe=None, w_tiles=None):
    """y = act(f16(epilogue(implicit_gemm(x, w))) (+ residual)) -- see fp_igemm_f16_fwd.  conv_rounding: nn.Conv2d under
    autocast (accumulator rounded to fp16 before the bias add, optional BatchNorm as scale/shift with its own rounding);
    otherwise nn.Linear (one rounding of accumulator + bias).  pe (S, N) f32 + y_pe (M, N) fp16: second output
    f16(f32(y) + pe[m % S]).  All tensors are device buffers owned by the caller (y is written in place and returned)."""
    x = _dev(x, torch.float16, "x"); w = _dev(w, torch.float16, "w"); y = _dev(y, torch.float16, "y")
    b = _dev(bias, torch.float32, "bias"); r = _dev(residual, torch.float16, "residual")
    sc = _dev(bn_scale, torch.float32, "bn_scale"); sh = _dev(bn_shift, torch.float32, "bn_shift")
    pe = _dev(pe, torch.float32, "pe"); y_pe = _dev(y_pe, torch.float16, "y_pe")
    ep = IgemmEpilogue()
    ep.bias, ep.bn_scale, ep.bn_shift = _ptr(b), _ptr(sc), _ptr(sh)
    ep.residual = _ptr(r)
    ep.r_geom = C.pointer(r_geom) if r_geom is not None else None
    ep.flags = (IGEMM_RELU if relu else 0) | (IGEMM_ROUND_ACC if conv_rounding else 0) | IGEMM_HAS_W_TILES
    ep.pe, ep.pe_period, ep.y_pe = _ptr(pe), (int(pe.shape[-2]) if pe is not None else 0), _ptr(y_pe)
    ep.w_tiles = _ptr(_dev(w_tiles, torch.float16, "w_tiles"))
    st = _lib.lib().fp_igemm_f16_fwd(_ptr(x), C.byref(x_geom), _ptr(w), _ptr(y), C.byref(y_geom), int(M), int(N), int(Cin),
                                     int(taps), C.byref(ep), _stream(x))
    _lib.check(st, "fp_igemm_f16_fwd")
    return y


def igemm_splitk_workspace_bytes(M, N, splits):
    return int(_lib.lib().fp_igemm_splitk_workspace_bytes(int(M), int(N), int(splits)))


def igemm_f16_splitk(x, x_geom, w, bias, y, y_geom, M, N, Cin, taps, splits, workspace, relu=False, residual=None, r_geom=None,
                     bn_scale=None, bn_shift=None, conv_rounding=False, pe=None, y_pe=None):
    """igemm_f16 for launches of a few dozen tiles (one or two hypotheses: the reference's track_one): the k range in `splits`
    pieces, partial sums through the caller-owned `workspace` (uint8, >= igemm_splitk_workspace_bytes), see
    fp_igemm_f16_splitk_fwd.  Equal to igemm_f16 up to fp32 summation order."""
    x = _dev(x, torch.float16, "x"); w = _dev(w, torch.float16, "w"); y = _dev(y, torch.float16, "y")
    b = _dev(bias, torch.float32, "bias"); r = _dev(residual, torch.float16, "residual")
    sc = _dev(bn_scale, torch.float32, "bn_scale"); sh = _dev(bn_shift, torch.float32, "bn_shift")
    pe = _dev(pe, torch.float32, "pe"); y_pe = _dev(y_pe, torch.float16, "y_pe")
    ws = _dev(workspace, torch.uint8, "workspace")
    ep = IgemmEpilogue()
    ep.bias, ep.bn_scale, ep.bn_shift = _ptr(b), _ptr(sc), _ptr(sh)
    ep.residual = _ptr(r)
    ep.r_geom = C.pointer(r_geom) if r_geom is not None else None
    ep.flags = (IGEMM_RELU if relu else 0) | (IGEMM_ROUND_ACC if conv_rounding else 0)
    ep.pe, ep.pe_period, ep.y_pe = _ptr(pe), (int(pe.shape[-2]) if pe is not None else 0), _ptr(y_pe)
    st = _lib.lib().fp_igemm_f16_splitk_fwd(_ptr(x), C.byref(x_geom), _ptr(w), _ptr(y), C.byref(y_geom), int(M), int(N), int(Cin),
                                            int(taps), C.byref(ep), int(splits), _ptr(ws), ws.numel(), _stream(x))
    _lib.check(st, "fp_igemm_f16_splitk_fwd")
    return y


def _work_igemm_splitk(x, x_geom, w, bias, y, y_geom, M, N, Cin, taps, splits, workspace, relu=False, residual=None, r_geom=None, **k):
    return _work_igemm(x, x_geom, w, bias, y, y_geom, M, N, Cin, taps, relu=relu, residual=residual, r_geom=r_geom)


def _work_igemm(x, x_geom, w, bias, y, y_geom, M, N, Cin, taps, relu=False, residual=None, r_geom=None, **k):
    by = 2 * (M * Cin * (1 if taps == 1 else 1.0 / (x_geom.stride ** 2)) + N * Cin * taps + M * N * (2 if residual is not None else 1))
    return by, 2.0 * M * N * Cin * taps


def add_pe_f16(tok, pe):
    """tok (B, S, 512) fp16, pe (S, 512) f32 -> f16(f32(tok) + pe): the in_proj operand (fp_add_pe_f16_fwd)"""
    tok = _dev(tok, torch.float16, "tok")
    pe = _dev(pe, torch.float32, "pe")
    S, D = int(pe.shape[-2]), int(pe.shape[-1])
    M = tok.numel() // D
    out = torch.empty_like(tok)
    _lib.check(_lib.lib().fp_add_pe_f16_fwd(_ptr(tok), _ptr(pe), _ptr(out), M, S, D, _stream(tok)), "fp_add_pe_f16_fwd")
    return out


def replicate_channels(buf, n, c0, c1):
    """buf (>= n, Hp, Wp, C) fp16 NHWC, contiguous: buf[1:n, :, :, c0:c1] = buf[0, :, :, c0:c1] (fp_replicate_rows_f16)"""
    buf = _dev(buf, torch.float16, "buf")
    if buf.dim() != 4 or not buf.is_contiguous() or buf.shape[0] < n:
        raise _lib.FpAmdError(f"replicate_channels: buf must be a contiguous (>= {n}, Hp, Wp, C) tensor, got {tuple(buf.shape)}")
    if n <= 1:
        return buf
    _, Hp, Wp, Ct = (int(v) for v in buf.shape)
    src = buf.data_ptr() + 2 * int(c0)
    st = _lib.lib().fp_replicate_rows_f16(C.c_void_p(src), C.c_void_p(src + 2 * Hp * Wp * Ct), int(n) - 1, Hp * Wp, int(c1) - int(c0), Ct, Ct,
                                          Hp * Wp * Ct, _stream(buf))
    _lib.check(st, "fp_replicate_rows_f16")
    return buf


def layernorm_res(branch16, gamma, beta, eps=1e-5, x32=None, tok16=None, pe=None, want32=True, want16=True):
    """LN(resid + f32(branch16)) * gamma + beta with resid = x32 or f32(tok16) + pe -> (y32 | None, y16 | None)
    (fp_layernorm_res_fwd: the fp32 residual stream / LayerNorms of nn.TransformerEncoderLayer under autocast)"""
    br = _dev(branch16, torch.float16, "branch16")
    D = int(br.shape[-1])
    M = br.numel() // D
    x32 = _dev(x32, torch.float32, "x32"); tok16 = _dev(tok16, torch.float16, "tok16"); pe = _dev(pe, torch.float32, "pe")
    S = int(pe.shape[-2]) if pe is not None else 0
    y32 = torch.empty(br.shape, dtype=torch.float32, device=br.device) if want32 else None
    y16 = torch.empty_like(br) if want16 else None
    st = _lib.lib().fp_layernorm_res_fwd(_ptr(x32), _ptr(tok16), _ptr(pe), S, _ptr(br), _ptr(_dev(gamma, torch.float32, "gamma")),
                                         _ptr(_dev(beta, torch.float32, "beta")), float(eps), _ptr(y32), _ptr(y16), M, D, _stream(br))
    _lib.check(st, "fp_layernorm_res_fwd")
    return y32, y16


class PackedLinear512:
    """fragment-packed copy of a (512 n, 512) fp16 nn.Linear weight (fp_pack_linear512_f16 per block of 512 output channels): what
    linear512, linear_layernorm_res and ffn_layernorm_mean take, whose waves read their weight rows straight from L2 into MFMA
    operand registers"""

    def __init__(self, w16):
        w = _dev(w16, torch.float16, "w16")
        if w.dim() != 2 or int(w.shape[1]) != 512 or int(w.shape[0]) % 512 or int(w.shape[0]) == 0:
            raise _lib.FpAmdError(f"PackedLinear512: weight {tuple(w.shape)}, must be (512 n, 512)")
        self.out_features = int(w.shape[0])
        self.data = torch.empty_like(w)
        for blk in range(self.out_features // 512):
            _lib.check(_lib.lib().fp_pack_linear512_f16(_ptr(w[blk * 512:]), _ptr(self.data[blk * 512:]), _stream(w)), "fp_pack_linear512_f16")


def _packed(w, name, out_features=512):
    if not isinstance(w, PackedLinear512):
        raise _lib.FpAmdError(f"{name}: the weight must be a PackedLinear512 (ops.PackedLinear512(w16))")
    if out_features is not None and w.out_features != out_features:
        raise _lib.FpAmdError(f"{name}: packed weight has {w.out_features} output features, expected {out_features}")
    return w.data


def linear512(x16, w_packed, bias, relu=False, out=None):
    """f16(x16 @ W^T + bias) for x16 (..., 512) fp16 and a PackedLinear512 of W (512 n, 512), n <= 6 (fp_linear512_f16_fwd): the
    input tile of a workgroup is fetched once for all n column blocks; the bits of igemm_f16 with taps = 1"""
    x = _dev(x16, torch.float16, "x16")
    w = _packed(w_packed, "linear512", None)
    if int(x.shape[-1]) != 512:
        raise _lib.FpAmdError(f"linear512: x16 (..., {int(x.shape[-1])}), must be (..., 512)")
    N = w_packed.out_features
    M = x.numel() // 512
    y = out if out is not None else torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float16, device=x.device)
    if y.dtype != torch.float16 or y.numel() != M * N or not y.is_contiguous():
        raise _lib.FpAmdError("linear512: out must be a contiguous fp16 tensor of M x N elements")
    _lib.check(_lib.lib().fp_linear512_f16_fwd(_ptr(x), _ptr(w), _ptr(_dev(bias, torch.float32, "bias")), _ptr(y), M, N, 1 if relu else 0,
                                               _stream(x)), "fp_linear512_f16_fwd")
    return y


def linear_layernorm_res(x16, w_packed, bias, gamma, beta, eps=1e-5, x32=None, tok16=None, pe=None, want32=True, want16=True):
    """layernorm_res(f16(x16 @ w^T + bias), ...) in ONE launch, the product staying on chip (fp_linear_layernorm_fwd).
    x16 (..., 512) fp16, w_packed = PackedLinear512 of the (512, 512) weight -> (y32 | None, y16 | None) of shape (..., 512);
    bit-identical to the two-kernel path.  x16 may be a column block `wide[..., c:c + 512]` of a contiguous wider tensor (one head's
    half of a two-head attention output): the kernel reads it with the wide row stride"""
    w = _packed(w_packed, "linear_layernorm_res")
    if not (torch.is_tensor(x16) and x16.is_cuda and x16.dtype == torch.float16):
        raise _lib.FpAmdError("linear_layernorm_res: x16 must be a CUDA float16 tensor")
    K = int(x16.shape[-1])
    D = 512
    if K != 512:
        raise _lib.FpAmdError(f"linear_layernorm_res: x16 (..., {K}), must be (..., 512)")
    M = x16.numel() // K
    if x16.is_contiguous():
        x, ldx = x16, K
    else:
        # a column block of a contiguous (..., ldx) tensor: unit stride in the last dimension, every leading stride that of the wide tensor
        ldx = int(x16.stride(-2)) if x16.dim() >= 2 else K
        ok = x16.stride(-1) == 1 and ldx % 8 == 0 and ldx >= K and x16.storage_offset() % 8 == 0
        for d in range(x16.dim() - 2, 0, -1):
            ok = ok and x16.stride(d - 1) == x16.stride(d) * x16.shape[d]
        if not ok:
            raise _lib.FpAmdError("linear_layernorm_res: x16 must be contiguous or a column block of a contiguous tensor")
        x = x16
    x32 = _dev(x32, torch.float32, "x32"); tok16 = _dev(tok16, torch.float16, "tok16"); pe = _dev(pe, torch.float32, "pe")
    S = int(pe.shape[-2]) if pe is not None else 0
    shape = tuple(x.shape[:-1]) + (D,)
    y32 = torch.empty(shape, dtype=torch.float32, device=x.device) if want32 else None
    y16 = torch.empty(shape, dtype=torch.float16, device=x.device) if want16 else None
    st = _lib.lib().fp_linear_layernorm_fwd(_ptr(x), _ptr(w), _ptr(_dev(bias, torch.float32, "bias")), _ptr(x32), _ptr(tok16), _ptr(pe), S,
                                            _ptr(_dev(gamma, torch.float32, "gamma")), _ptr(_dev(beta, torch.float32, "beta")), float(eps),
                                            _ptr(y32), _ptr(y16), M, K, D, ldx, _stream(x))
    _lib.check(st, "fp_linear_layernorm_fwd")
    return y32, y16


def ffn_layernorm_mean(y16, w1_packed, b1, w2_packed, b2, x32, gamma, beta, eps=1e-5):
    """(G, R, 512) fp16 y16 (norm1's output) + (G, R, 512) f32 residual stream -> (G, 512) f32 =
    mean_r LN(x32 + linear2(relu(linear1(y16)))) * gamma + beta, one launch + a finish kernel (fp_ffn_layernorm_mean_fwd):
    the feed-forward half of the encoder layer and the token mean with both (G*R, 512) intermediates staying on chip;
    w1_packed / w2_packed: PackedLinear512 of the two (512, 512) weights"""
    y = _dev(y16, torch.float16, "y16")
    G_, R, D = (int(v) for v in y.shape)
    if D != 512:
        raise _lib.FpAmdError("ffn_layernorm_mean: d_model must be 512")
    x32 = _dev(x32, torch.float32, "x32")
    out = torch.empty((G_, D), dtype=torch.float32, device=y.device)
    if R % 16:
        raise _lib.FpAmdError(f"ffn_layernorm_mean: {R} rows per group, must be a multiple of 16")
    ws = torch.empty((max(G_ * R // 16, 1), 512), dtype=torch.float32, device=y.device)
    st = _lib.lib().fp_ffn_layernorm_mean_fwd(_ptr(y), _ptr(_packed(w1_packed, "ffn_layernorm_mean")), _ptr(_dev(b1, torch.float32, "b1")),
                                              _ptr(_packed(w2_packed, "ffn_layernorm_mean")), _ptr(_dev(b2, torch.float32, "b2")), _ptr(x32),
                                              _ptr(_dev(gamma, torch.float32, "gamma")), _ptr(_dev(beta, torch.float32, "beta")), float(eps),
                                              _ptr(out), _ptr(ws), ws.numel() * 4, G_, R, _stream(y))
    _lib.check(st, "fp_ffn_layernorm_mean_fwd")
    return out


def _column_block(x16, name):
    """-> row stride (fp16 values) of an (..., 512) fp16 tensor that is contiguous or a column block of a contiguous wider tensor"""
    if not (torch.is_tensor(x16) and x16.is_cuda and x16.dtype == torch.float16 and int(x16.shape[-1]) == 512):
        raise _lib.FpAmdError(f"{name}: must be a CUDA float16 tensor (..., 512)")
    if x16.is_contiguous():
        return 512
    ldx = int(x16.stride(-2)) if x16.dim() >= 2 else 512
    ok = x16.stride(-1) == 1 and ldx % 8 == 0 and ldx >= 512 and x16.storage_offset() % 8 == 0
    for d in range(x16.dim() - 2, 0, -1):
        ok = ok and x16.stride(d - 1) == x16.stride(d) * x16.shape[d]
    if not ok:
        raise _lib.FpAmdError(f"{name}: must be contiguous or a column block of a contiguous tensor")
    return ldx


_TAIL_WS = {}


def encoder_tail_mean(ctx16, wo_packed, bo, tok16, pe, gamma1, beta1, w1_packed, b1, w2_packed, b2, gamma2, beta2, eps=1e-5, workspace=None):
    """(G, R, 512) fp16 attention context (heads merged; may be a column block of a wider tensor) + the layer input f32(tok16) + pe ->
    (G, 512) f32 = mean_r LN2(y + linear2(relu(linear1(f16(y))))), y = LN1(f32(tok16) + pe + f16(ctx16 @ Wo^T + bo)): everything of the
    encoder layer behind the attention + the token mean in one launch (fp_encoder_tail_mean_fwd) = linear_layernorm_res followed by
    ffn_layernorm_mean, bit for bit, without norm1's fp16 output reaching HBM"""
    ldx = _column_block(ctx16, "encoder_tail_mean: ctx16")
    G_, R, _ = (int(v) for v in ctx16.shape)
    tok = _dev(tok16, torch.float16, "tok16")
    pe = _dev(pe, torch.float32, "pe")
    if int(pe.shape[-2]) != R or tuple(tok.shape) != (G_, R, 512):
        raise _lib.FpAmdError("encoder_tail_mean: tok16 must be (G, R, 512) and pe (R, 512)")
    if R % 16:
        raise _lib.FpAmdError(f"encoder_tail_mean: {R} rows per group, must be a multiple of 16")
    out = torch.empty((G_, 512), dtype=torch.float32, device=ctx16.device)
    need = int(_lib.lib().fp_encoder_tail_workspace_bytes(G_, R))
    ws = workspace if workspace is not None else torch.empty(need, dtype=torch.uint8, device=ctx16.device)
    f32 = lambda t, n: _ptr(_dev(t, torch.float32, n))
    st = _lib.lib().fp_encoder_tail_mean_fwd(_ptr(ctx16), ldx, _ptr(_packed(wo_packed, "encoder_tail_mean")), f32(bo, "bo"), _ptr(tok), _ptr(pe),
                                             f32(gamma1, "gamma1"), f32(beta1, "beta1"), _ptr(_packed(w1_packed, "encoder_tail_mean")), f32(b1, "b1"),
                                             _ptr(_packed(w2_packed, "encoder_tail_mean")), f32(b2, "b2"), f32(gamma2, "gamma2"), f32(beta2, "beta2"),
                                             float(eps), _ptr(out), _ptr(ws), ws.numel(), G_, R, _stream(ctx16))
    _lib.check(st, "fp_encoder_tail_mean_fwd")
    return out


def colmean_f16(x, gamma=None, beta=None, eps=1e-5, resid32=None):
    """x (G, R, 512) fp16 -> (G, 512) f32: mean over R of LN(resid32 + x)*gamma+beta (gamma given) or of x (fp_colmean_f16_fwd)"""
    x = _dev(x, torch.float16, "x")
    G_, R, D = (int(v) for v in x.shape)
    out = torch.empty((G_, D), dtype=torch.float32, device=x.device)
    st = _lib.lib().fp_colmean_f16_fwd(_ptr(x), _ptr(_dev(resid32, torch.float32, "resid32")), _ptr(_dev(gamma, torch.float32, "gamma")),
                                       _ptr(_dev(beta, torch.float32, "beta")), float(eps), _ptr(out), G_, R, D, _stream(x))
    _lib.check(st, "fp_colmean_f16_fwd")
    return out


ROWS_ROUND_F16, ROWS_X_F16, ROWS_Y_F16 = 1, 2, 4


def rows_linear(x, w, bias=None, round_f16=False, out_f16=False, out=None):
    """y = x @ w.T + bias for a few hundred rows: x (M,K) f32|f16, w (N,K) f16, bias (N) f32 -> (M,N) f32 (rounded to
    fp16 values if round_f16) or fp16 (out_f16) (fp_rows_linear_fwd).  out: optional (M,N) destination of that dtype"""
    if not (torch.is_tensor(x) and x.dtype in (torch.float32, torch.float16)):
        raise _lib.FpAmdError("rows_linear: x must be an f32 or f16 tensor")
    x = _dev(x, x.dtype, "x"); w = _dev(w, torch.float16, "w"); b = _dev(bias, torch.float32, "bias")
    M, K = (int(v) for v in x.shape)
    N = int(w.shape[0])
    ydt = torch.float16 if out_f16 else torch.float32
    y = torch.empty((M, N), dtype=ydt, device=x.device) if out is None else _dev(out, ydt, "out")
    if tuple(y.shape) != (M, N):
        raise _lib.FpAmdError(f"rows_linear: out has shape {tuple(y.shape)}, expected {(M, N)}")
    flags = (ROWS_ROUND_F16 if round_f16 else 0) | (ROWS_X_F16 if x.dtype == torch.float16 else 0) | (ROWS_Y_F16 if out_f16 else 0)
    _lib.check(_lib.lib().fp_rows_linear_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, K, N, flags, _stream(x)), "fp_rows_linear_fwd")
    return y


ATT_FP16_SCORES = 1


def attention_f16(qkv, n_heads, fp16_scores=False):
    """qkv (B, S, 3*D) fp16 = in_proj output [q | k | v] -> (B, S, D) fp16 = softmax(q k^T / sqrt(hd)) v, heads merged
    (fp_attention_f16_fwd; head size D / n_heads must be 128).  fp16_scores: q * sqrt(1/hd) and the scores rounded to fp16
    (the need_weights=True branch of nn.MultiheadAttention under autocast, score_network.py:73,86)"""
    qkv = _dev(qkv, torch.float16, "qkv")
    B, S, D3 = (int(v) for v in qkv.shape)
    D = D3 // 3
    out = torch.empty((B, S, D), dtype=torch.float16, device=qkv.device)
    st = _lib.lib().fp_attention_f16_fwd(_ptr(qkv), _ptr(out), B, S, int(n_heads), D // int(n_heads),
                                         ATT_FP16_SCORES if fp16_scores else 0, _stream(qkv))
    _lib.check(st, "fp_attention_f16_fwd")
    return out


def cluster_poses(angle_diff, dist_diff, poses, symmetry_tfs):
    """Host op (init-time): returns indices of the kept poses (mycpp.cluster_poses semantics)."""
    P = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).reshape(-1, 16))
    S = np.ascontiguousarray(np.asarray(symmetry_tfs, dtype=np.float32).reshape(-1, 16))
    keep = np.empty(P.shape[0], np.int32)
    n = _lib.lib().fp_cluster_poses(float(angle_diff), float(dist_diff), P.ctypes.data_as(C.c_void_p), P.shape[0],
                                    S.ctypes.data_as(C.c_void_p), S.shape[0], keep.ctypes.data_as(C.c_void_p))
    return keep[:n].copy()


# ---------------------------------------------------------------------------------------------------------------
# optional per-entry-point timing with HIP events on the launch stream (used by bench.py for the roofline numbers)
class KernelTimers:
    """with KernelTimers() as t: ...; t.summary() -> {name: dict(calls, avg_ms, bytes, flops)}; times come from
    HIP events recorded on the stream the kernels are launched on; bytes/flops are the ALGORITHMIC work of one
    launch (DESIGN.md 'Kernels'), averaged over the recorded launches."""

    active = None

    def __init__(self):
        self.records = {}

    def __enter__(self):
        KernelTimers.active = self
        return self

    def __exit__(self, *a):
        KernelTimers.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in evs]
            n = max(1, len(ms))
            out[name] = dict(calls=len(ms), avg_ms=float(sum(ms) / n), bytes=float(sum(e[2] for e in evs) / n),
                             flops=float(sum(e[3] for e in evs) / n))
        return out

    def busy(self, ref):
        """for launches recorded on SEVERAL streams: per entry point the BUSY time = length of the union of its launches'
        [start, end] intervals (timestamps relative to the event `ref`, recorded before the first launch), next to the sum of
        the launch durations, total algorithmic bytes / flops and the span first start .. last end.
        -> {name: dict(calls, sum_ms, busy_ms, bytes, flops, first_ms, last_ms)}"""
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.records.items():
            iv = sorted((ref.elapsed_time(a), ref.elapsed_time(b)) for a, b, _, _ in evs)
            busy, cs, ce = 0.0, None, None
            for a, b in iv:
                if cs is None:
                    cs, ce = a, b
                elif a <= ce:
                    ce = max(ce, b)
                else:
                    busy += ce - cs
                    cs, ce = a, b
            if cs is not None:
                busy += ce - cs
            out[name] = dict(calls=len(iv), sum_ms=float(sum(b - a for a, b in iv)), busy_ms=float(busy),
                             bytes=float(sum(e[2] for e in evs)), flops=float(sum(e[3] for e in evs)),
                             first_ms=float(iv[0][0]) if iv else 0.0, last_ms=float(max(b for _, b in iv)) if iv else 0.0)
        return out


def _esz(t):
    return t.element_size() if t is not None else 4


def _work_render(mesh, poses, bbox2d, K, H, W, out_hw=(160, 160), *a, **k):
    N = int(poses.shape[0])
    A = k.get("A_out")
    esz = _esz(A) if A is not None else (2 if k.get("out_f16") else 4)
    return N * (6 * out_hw[0] * out_hw[1] * esz + 32 * mesh.V + 12 * mesh.T), 0.0


def _work_warp(rgb, xyz_map, depth, tf_to_crops, K, poses, *a, **k):
    N = int(poses.shape[0])
    B = k.get("B_out")
    oh, ow = k.get("out_hw", (160, 160))
    esz = _esz(B) if B is not None else (2 if k.get("out_f16") else 4)
    return N * 6 * oh * ow * esz + rgb.shape[0] * rgb.shape[1] * 24, 0.0


def _work_conv1(x, *a, **k):
    Bn, _, H, W = x.shape
    return Bn * (6 * H * W + 64 * (H // 2) * (W // 2)) * 2 + 64 * 294 * 2, 2.0 * Bn * (H // 2) * (W // 2) * 64 * 294


def _timed(name, fn, work=None):
    def wrapper(*a, **k):
        t = KernelTimers.active
        if t is None:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        by, fl = work(*a, **k) if work is not None else (0.0, 0.0)
        t.records.setdefault(name, []).append((e0, e1, float(by), float(fl)))
        return r
    wrapper.__name__ = fn.__name__
    wrapper.__doc__ = fn.__doc__
    return wrapper


render_crops = _timed("fp_render_crops", render_crops, _work_render)
warp_crops = _timed("fp_warp_crops", warp_crops, _work_warp)
crop_windows = _timed("fp_crop_windows", crop_windows)
pose_update = _timed("fp_pose_update", pose_update)
conv7x7s2_bn_relu = _timed("fp_conv7x7s2_bn_relu_fwd", conv7x7s2_bn_relu, _work_conv1)
igemm_f16 = _timed("fp_igemm_f16_fwd", igemm_f16, _work_igemm)
igemm_f16_splitk = _timed("fp_igemm_f16_splitk_fwd", igemm_f16_splitk, _work_igemm_splitk)
add_pe_f16 = _timed("fp_add_pe_f16_fwd", add_pe_f16, lambda tok, pe: (4.0 * tok.numel(), 0.0))
replicate_channels = _timed("fp_replicate_rows_f16", replicate_channels,
                            lambda buf, n, c0, c1: (2.0 * n * buf.shape[1] * buf.shape[2] * (c1 - c0), 0.0))
layernorm_res = _timed("fp_layernorm_res_fwd", layernorm_res,
                       lambda br, *a, **k: ((2.0 + (4.0 if k.get("x32") is not None else 2.0) + (4.0 if k.get("want32", True) else 0.0)
                                             + (2.0 if k.get("want16", True) else 0.0)) * br.numel(), 0.0))
linear512 = _timed("fp_linear512_f16_fwd", linear512,
                   lambda x, w, *a, **k: (2.0 * x.numel() + 2.0 * 512 * w.out_features + 2.0 * (x.numel() // 512) * w.out_features,
                                          2.0 * x.numel() * w.out_features))
linear_layernorm_res = _timed("fp_linear_layernorm_fwd", linear_layernorm_res,
                              lambda x, w, *a, **k: (2.0 * x.numel() + 2.0 * 512 * 512 + (x.numel() // x.shape[-1]) * 512 *
                                                     ((4.0 if k.get("x32") is not None else 2.0) + (4.0 if k.get("want32", True) else 0.0)
                                                      + (2.0 if k.get("want16", True) else 0.0)),
                                                     2.0 * x.numel() * 512))
ffn_layernorm_mean = _timed("fp_ffn_layernorm_mean_fwd", ffn_layernorm_mean,
                            lambda y, w1, b1, w2, *a, **k: (6.0 * y.numel() + 4.0 * 512 * 512, 4.0 * y.numel() * 512))
encoder_tail_mean = _timed("fp_encoder_tail_mean_fwd", encoder_tail_mean,
                           lambda ctx, *a, **k: (2.0 * ctx.numel() * 2 + 8.0 * ctx.numel() + 6.0 * 512 * 512, 6.0 * ctx.numel() * 512))
colmean_f16 = _timed("fp_colmean_f16_fwd", colmean_f16,
                     lambda x, *a, **k: ((6.0 if k.get("resid32") is not None else 2.0) * x.numel(), 0.0))
rows_linear = _timed("fp_rows_linear_fwd", rows_linear)
attention_f16 = _timed("fp_attention_f16_fwd", attention_f16,
                       lambda qkv, n_heads, **k: (2.0 * qkv.numel() * 4.0 / 3.0, 4.0 * qkv.shape[0] * qkv.shape[1] ** 2 * (qkv.shape[2] // 3)))
