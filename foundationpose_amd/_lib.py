"""ctypes binding of libfp_amd.so (the C ABI declared in include/fp_amd.h).

Fails loudly: there is no fallback path.  torch is imported first so that the HIP runtime
already mapped by PyTorch-ROCm (soname libamdhip64.so.7) is the one our library binds to.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime with PyTorch)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FP_AMD_LIB") or os.path.join(_HERE, "csrc", "libfp_amd.so")   # FP_AMD_LIB: A/B builds
ABI_VERSION = 213    # = FP_AMD_ABI_VERSION of include/fp_amd.h (tests/test_abi.py keeps the two in step)
_lib = None

vp, ci, cf, cd, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/fp_amd.h (checked by tests/test_abi.py)
SIGNATURES = {
    "fp_last_error": (C.c_char_p, []),
    "fp_version": (ci, []),
    "fp_mesh_create": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, C.POINTER(vp)]),
    "fp_mesh_destroy": (None, [vp]),
    "fp_depth_erode": (ci, [vp, vp, ci, ci, ci, cf, cf, cf, vp]),
    "fp_depth_bilateral": (ci, [vp, vp, ci, ci, ci, cf, cf, cf, vp]),
    "fp_depth_to_xyz": (ci, [vp, vp, cf, ci, vp, ci, ci, vp]),
    "fp_crop_windows": (ci, [vp, vp, cd, cd, ci, ci, ci, vp, vp, vp]),
    "fp_workspace_bytes": (sz, [ci, ci, ci, ci, ci]),
    "fp_render_crops": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, cf, cf, cf, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "fp_warp_crops": (ci, [vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "fp_pose_update": (ci, [vp, vp, vp, ci, ci, vp, cf, cf, ci, vp, vp, vp, ci, vp, vp, cf, vp]),
    "fp_conv7x7s2_bn_relu_fwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "fp_igemm_f16_fwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "fp_pack_conv3x3_tiles_f16": (ci, [vp, vp, ci, ci, vp]),
    "fp_igemm_splitk_workspace_bytes": (sz, [ci, ci, ci]),
    "fp_igemm_f16_splitk_fwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, ci, vp, sz, vp]),
    "fp_add_pe_f16_fwd": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "fp_replicate_rows_f16": (ci, [vp, vp, ci, ci, ci, ci, ci, C.c_longlong, vp]),
    "fp_layernorm_res_fwd": (ci, [vp, vp, vp, ci, vp, vp, vp, cf, vp, vp, ci, ci, vp]),
    "fp_pack_linear512_f16": (ci, [vp, vp, vp]),
    "fp_linear512_f16_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, vp]),
    "fp_linear_layernorm_fwd": (ci, [vp, vp, vp, vp, vp, vp, ci, vp, vp, cf, vp, vp, ci, ci, ci, ci, vp]),
    "fp_ffn_layernorm_mean_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, cf, vp, vp, C.c_size_t, ci, ci, vp]),
    "fp_encoder_tail_workspace_bytes": (sz, [ci, ci]),
    "fp_encoder_tail_mean_fwd": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, vp, vp, sz, ci, ci, vp]),
    "fp_colmean_f16_fwd": (ci, [vp, vp, vp, vp, cf, vp, ci, ci, ci, vp]),
    "fp_rows_linear_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "fp_attention_f16_fwd": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "fp_cluster_poses": (ci, [cf, cf, vp, ci, vp, ci, vp]),
}


class FpAmdError(RuntimeError):
    pass


def lib():
    """Load libfp_amd.so or raise (build it with `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FpAmdError(
                f"{LIB_PATH} is missing: the HIP extension is not built (run __graft_entry__.build() or "
                f"`make -C foundationpose_amd/csrc`). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is absent: loud by design
            fn.restype = res
            fn.argtypes = args
        got = int(handle.fp_version())
        if got != ABI_VERSION:
            raise FpAmdError(f"{LIB_PATH} reports ABI version {got}, this binding was written for {ABI_VERSION}: rebuild the library "
                             f"(`make -C foundationpose_amd/csrc`) -- entry points changed meaning between versions (include/fp_amd.h)")
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().fp_last_error()
        raise FpAmdError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")
