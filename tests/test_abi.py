"""CPU: the C-ABI library loads and exports exactly what include/fp_amd.h declares."""
import os
import re

from conftest import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "fp_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from foundationpose_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 15
    lib = _lib.lib()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in fp_amd.h but not exported by libfp_amd.so"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in foundationpose_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    # the ABI version: header, library and binding agree (round 5: fp_linear_layernorm_fwd changed the meaning of its weight argument
    # under an unchanged name -- an external caller can only notice through the version)
    import re
    hdr = open(os.path.join(ROOT, "include", "fp_amd.h")).read()
    ver = int(re.search(r"#define\s+FP_AMD_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.fp_version() == ver == _lib.ABI_VERSION and ver >= 210


def test_argument_errors_are_reported_without_gpu():
    import ctypes as C
    from foundationpose_amd import _lib
    lib = _lib.lib()
    h = C.c_void_p()
    st = lib.fp_mesh_create(None, None, None, None, None, None, None, 0, 0, 0, 0, C.byref(h))
    assert st == -1 and b"fp_mesh_create" in lib.fp_last_error()
    assert lib.fp_rows_linear_fwd(C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), 4, 33, 128, 0, None) == -1
    assert b"multiple" in lib.fp_last_error()
    # scratch = per-hypothesis vertex records (32 B / vertex) + per-strip triangle lists (10 strips of 16 rows)
    ws = lib.fp_workspace_bytes(252, 2501, 4900, 160, 160)
    assert 252 * (2501 * 32 + 10 * 4900 * 2) <= ws <= 252 * (2501 * 32 + 10 * 4900 * 2 + 10 * 4) + 4096
    assert lib.fp_workspace_bytes(0, 2501, 4900, 160, 160) == 0
    big = lib.fp_workspace_bytes(4, 100000, 200000, 160, 160)        # > 65535 triangles: 32-bit ids in the lists
    assert big >= 4 * (100000 * 32 + 10 * 200000 * 4)
    # GEMM geometry / epilogue errors
    from foundationpose_amd.ops import IgemmEpilogue
    G = (C.c_int * 10)(1, 1, 1, 1, 1, 0, 512, 0, 0, 0)
    assert lib.fp_igemm_f16_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 100, 512, 1, None, None) == -1
    assert b"multiple of 128" in lib.fp_last_error()
    ep = IgemmEpilogue()
    ep.bn_scale, ep.bn_shift, ep.flags = 16, 16, 1      # BatchNorm in the epilogue only with the conv rounding sequence
    assert lib.fp_igemm_f16_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 128, 512, 1, C.byref(ep), None) == -1
    assert b"FP_IGEMM_ROUND_ACC" in lib.fp_last_error()
    ep = IgemmEpilogue()
    ep.flags = 8                                        # unknown flags are refused
    assert lib.fp_igemm_f16_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 128, 512, 1, C.byref(ep), None) == -1
    ep = IgemmEpilogue()
    ep.pe = 16                                          # positional table without its output / period
    assert lib.fp_igemm_f16_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 128, 512, 1, C.byref(ep), None) == -1
    # tile-packed conv weights: shape / alias checks of the repack, and w_tiles is refused for a plain GEMM
    assert lib.fp_pack_conv3x3_tiles_f16(C.c_void_p(16), C.c_void_p(16), 128, 64, None) == -1          # in place
    assert lib.fp_pack_conv3x3_tiles_f16(C.c_void_p(16), C.c_void_p(32), 100, 64, None) == -1 and b"multiple of 128" in lib.fp_last_error()
    ep = IgemmEpilogue()
    ep.w_tiles, ep.flags = 32, 4                             # FP_IGEMM_HAS_W_TILES: the member is read
    assert lib.fp_igemm_f16_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 128, 512, 1, C.byref(ep), None) == -1
    assert b"w_tiles" in lib.fp_last_error()
    # ... and the split-K entry point refuses it under ITS name instead of ignoring it (round 6; error strings carry the entry point)
    G9 = (C.c_int * 10)(1600, 40, 42, 42, 1, 0, 512, 0, 0, 0)
    assert lib.fp_igemm_f16_splitk_fwd(C.c_void_p(16), G9, C.c_void_p(16), C.c_void_p(16), G9, 4, 128, 512, 9, C.byref(ep), 4, C.c_void_p(16), 1 << 24, None) == -1
    assert b"fp_igemm_f16_splitk_fwd" in lib.fp_last_error() and b"w_tiles" in lib.fp_last_error()
    assert lib.fp_igemm_f16_splitk_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 100, 512, 1, None, 4, C.c_void_p(16), 1 << 24, None) == -1
    assert lib.fp_last_error().startswith(b"fp_igemm_f16_splitk_fwd: N=100")
    # split-K variant: workspace size, piece count and workspace checks come before any launch
    assert lib.fp_igemm_splitk_workspace_bytes(400, 512, 12) == 12 * 4 * 4 * 65536 and lib.fp_igemm_splitk_workspace_bytes(400, 100, 2) == 0
    assert lib.fp_igemm_f16_splitk_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 128, 512, 1, None, 0, C.c_void_p(16), 1 << 20, None) == -1
    assert b"splits=0" in lib.fp_last_error()
    assert lib.fp_igemm_f16_splitk_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 128, 512, 1, None, 9, C.c_void_p(16), 1 << 20, None) == -1
    assert lib.fp_igemm_f16_splitk_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 4, 128, 512, 1, None, 4, C.c_void_p(16), 1000, None) != 0
    assert b"workspace" in lib.fp_last_error()
    assert lib.fp_igemm_f16_splitk_fwd(C.c_void_p(16), G, C.c_void_p(16), C.c_void_p(16), G, 0, 128, 512, 1, None, 4, None, 0, None) == 0   # nothing to do
    assert lib.fp_layernorm_res_fwd(None, C.c_void_p(16), C.c_void_p(16), 400, C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 1e-5,
                                    C.c_void_p(16), None, 4, 256, None) == -1
    assert lib.fp_layernorm_res_fwd(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 400, C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 1e-5,
                                    C.c_void_p(16), None, 4, 512, None) == -1     # residual given twice
    assert lib.fp_attention_f16_fwd(C.c_void_p(16), C.c_void_p(16), 1, 4, 4, 128, 2, None) == -1
    assert lib.fp_linear_layernorm_fwd(C.c_void_p(16), C.c_void_p(16), None, None, C.c_void_p(16), C.c_void_p(16), 400, C.c_void_p(16),
                                       C.c_void_p(16), 1e-5, C.c_void_p(16), None, 4, 256, 512, 0, None) == -1      # K != 512
    assert b"K=256" in lib.fp_last_error()
    assert lib.fp_pack_linear512_f16(C.c_void_p(16), C.c_void_p(16), None) == -1                                   # in place
    assert lib.fp_pack_linear512_f16(None, C.c_void_p(16), None) == -1
    assert lib.fp_linear512_f16_fwd(C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), 4, 768, 0, None) == -1        # N % 512
    assert b"multiple of 512" in lib.fp_last_error()
    assert lib.fp_linear512_f16_fwd(C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), 0, 1536, 0, None) == 0        # nothing to do
    assert lib.fp_linear_layernorm_fwd(C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 400,
                                       C.c_void_p(16), C.c_void_p(16), 1e-5, C.c_void_p(16), None, 4, 512, 512, 0, None) == -1   # residual twice
    assert lib.fp_linear_layernorm_fwd(C.c_void_p(16), C.c_void_p(16), None, None, C.c_void_p(16), C.c_void_p(16), 400, C.c_void_p(16),
                                       C.c_void_p(16), 1e-5, C.c_void_p(16), None, 4, 512, 512, 500, None) == -1    # row stride below K
    assert b"ldx=500" in lib.fp_last_error()
    assert lib.fp_linear512_f16_fwd(C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), 4, 3584, 0, None) == -1       # N > 3072
    assert lib.fp_replicate_rows_f16(C.c_void_p(16), C.c_void_p(32), 3, 10, 100, 256, 256, 2560, None) == -1      # 100 channels
    assert b"multiples of 8" in lib.fp_last_error()
    assert lib.fp_replicate_rows_f16(C.c_void_p(16), C.c_void_p(32), 3, 10, 128, 64, 256, 2560, None) == -1       # stride < channels
    assert lib.fp_replicate_rows_f16(C.c_void_p(16), C.c_void_p(32), 0, 10, 128, 256, 256, 2560, None) == 0       # nothing to do
    # fp_render_crops defines two flag bits; anything else (e.g. the phase-skip bits of the profiling build) is refused
    for bad in (0x10000, 0x80000, 4):
        assert lib.fp_render_crops(None, None, None, None, 480, 640, 0, 160, 160, 0.8, 0.5, 0.17, 0.001, 3 | bad, None, None, None, None,
                                   None, None, None, None, 0, None) == -1
        assert b"unknown flag bits" in lib.fp_last_error()
    assert lib.fp_render_crops(None, None, None, None, 480, 640, 0, 160, 160, 0.8, 0.5, 0.17, 0.001, 3, None, None, None, None,
                               None, None, None, None, 0, None) == 0       # N == 0: nothing to do


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "foundationpose_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from foundationpose_amd import _lib, ops
    with pytest.raises(_lib.FpAmdError):
        ops.erode_depth(torch.zeros(4, 4))


def test_cluster_poses_host_op(scene):
    import numpy as np
    from foundationpose_amd import ops
    from foundationpose_amd.Utils import symmetry_tfs_from_info
    from oracle import ops as oo
    grid = scene["poses"].copy()
    keep = ops.cluster_poses(30, 99999, grid, np.eye(4)[None])
    assert len(keep) == 252  # identity symmetry: all 252 survive 30 deg clustering
    sym = symmetry_tfs_from_info({"symmetries_continuous": [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]})
    assert sym.shape == (73, 4, 4)
    k2 = ops.cluster_poses(30, 99999, grid, sym)
    k2o = oo.cluster_poses(30, 99999, grid, sym)
    assert len(k2) < 252 and np.array_equal(k2, k2o)
    assert np.array_equal(keep, oo.cluster_poses(30, 99999, grid, np.eye(4)[None]))


def test_transform_batch_refuses_an_unfused_batch():
    """the dataset shims pass a batch through only if it carries the fused network-input buffer of this package's
    make_crop_data_batch; anything else would need the un-fused normalisation (h5_dataset.py:79-170), which is not here"""
    import pytest
    from foundationpose_amd.h5_dataset import PoseRefinePairH5Dataset, ScoreMultiPairH5Dataset
    from foundationpose_amd.pose_dataset import BatchPoseData
    for ds in (PoseRefinePairH5Dataset(cfg={}, h5_file="", mode="test"), ScoreMultiPairH5Dataset(cfg={}, mode="test")):
        with pytest.raises(RuntimeError, match="fused"):
            ds.transform_batch(BatchPoseData(), H_ori=480, W_ori=640, bound=1)
        b = BatchPoseData()
        b.AB = object()
        assert ds.transform_batch(b, H_ori=480, W_ori=640, bound=1) is b
