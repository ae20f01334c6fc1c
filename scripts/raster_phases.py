"""phase timing of fp_render_crops with the phase-skip flag bits that exist only in the profiling build:
    make -C foundationpose_amd/csrc profile && FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so python scripts/raster_phases.py
(the product library rejects these bits: FP_ERR_INVALID_ARG)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from foundationpose_amd import ops, _lib
dev = torch.device("cuda:0"); N = int(os.environ.get("FP_N", "252"))
sc = bench.build_scene(dev, 0, N); h = sc["gm"]["_handle"]
poses = torch.as_tensor(sc["poses"], device=dev)
tf, bb = ops.crop_windows(poses, sc["K"], sc["diameter"], 1.2, (160, 160))
A = torch.empty((N, 6, 160, 160), dtype=torch.float16, device=dev)
L = _lib.lib(); ws = torch.empty(L.fp_workspace_bytes(N, h.V, h.T, 160, 160), dtype=torch.uint8, device=dev)
K9 = np.ascontiguousarray(np.asarray(sc["K"], np.float64).reshape(9).astype(np.float32))
def run(flags):
    st = L.fp_render_crops(h.handle, poses.data_ptr(), bb.data_ptr(), K9.ctypes.data_as(C.c_void_p), 480, 640, N, 160, 160, 0.8, 0.5,
                           float(sc["diameter"]), 0.001, flags, A.data_ptr(), None, None, None, None, None, None, ws.data_ptr(), ws.numel(),
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
for name, fl in (("full", 3), ("no shading", 3 | 0x10000), ("no phase1", 3 | 0x20000), ("neither", 3 | 0x30000),
                 ("setup only (no cells), no shading", 3 | 0x50000), ("cells without atomics, no shading", 3 | 0x90000)):
    run(fl); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(fl)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
cnts = ws[:0]
import struct
# strip list lengths
Ln = ops._lib.lib()
print("workspace bytes", ws.numel())
