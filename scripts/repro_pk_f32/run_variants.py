"""GPU half of the packed-fp32 bisection (make_variants.py builds the code objects on the CPU):

    python scripts/repro_pk_f32/run_variants.py [launches]       # prints one JSON line per variant

For every scripts/repro_pk_f32/variants/*.hsaco: the library's fp_render_crops runs once (k_vertex, k_bin and the PRODUCT
k_raster fill the workspace and give the reference tensors); then the variant's k_raster is launched from the loaded module on
the same workspace -- once alone (must equal the reference bit for bit) and `launches` times while a rocBLAS GEMM runs on
another stream (the condition under which the packed build returned wrong lanes 48-63).  Reported per variant:
launches that differ, and for the first differing launch which pixels / lanes / channels."""
import ctypes as C
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from foundationpose_amd import ops, synthetic as syn
from foundationpose_amd.Utils import get_mesh_handle, make_mesh_tensors
from foundationpose_amd.mesh import make_can_mesh

KERNEL = b"_Z8k_raster7fp_meshPKfS1_5fp_k9iiiiiffffi9RenderOut8RenderWs"
hip = C.CDLL("libamdhip64.so")
hip.hipModuleLoadData.argtypes = [C.POINTER(C.c_void_p), C.c_void_p]
hip.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: hipError {rc}")


def align256(x):
    return (x + 255) & ~255


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    mesh = make_can_mesh()
    gm = make_mesh_tensors(mesh, device=dev)
    h = get_mesh_handle(gm)
    n, oh, ow, H, W = 38, 160, 160, 480, 640
    T = syn.gt_pose(0).astype(np.float32)
    P = torch.as_tensor(syn.perturbed_poses(T, n, seed=3, max_trans=0.01, max_rot_deg=170.0).astype(np.float32), device=dev)
    diam = float(np.linalg.norm(mesh.vertices.max(0) - mesh.vertices.min(0)))
    K = syn.YCBV_K
    _, bb = ops.crop_windows(P, K, diam, 1.2, (ow, oh))
    ws = torch.empty(ops.workspace_bytes(n, h.V, h.T, oh, ow), dtype=torch.uint8, device=dev)
    A = torch.zeros((n, 6, oh, ow), dtype=torch.float16, device=dev)

    def product():
        return ops.render_crops(h, P, bb, K, H, W, out_hw=(oh, ow), mesh_diameter=diam, xyz_thr=0.001, normalize_xyz=True, A_out=A,
                                workspace=ws, want=("A", "zbuf", "tri_id"))
    ref = {k: v.clone() for k, v in product().items()}
    torch.cuda.synchronize()
    # ---- kernel arguments of k_raster (offsets from the code object's metadata: fp_mesh 0, poses 72, bbox 80, K 88, H 124, W 128,
    # oh 132, ow 136, nstrips 140, w_ambient 144, w_diffuse 148, inv_r 152, xyz_thr 156, flags 160, RenderOut 168, RenderWs 224)
    nstrips = (oh + 15) // 16
    V, Tn = h.V, h.T
    o_vr = 0
    o_va = align256(o_vr + n * V * 8)
    o_cnt = align256(o_va + n * V * 24)
    o_lst = align256(o_cnt + n * nstrips * 4)
    assert align256(o_lst + n * nstrips * Tn * 2) == ws.numel() and Tn <= 65535
    outA = torch.zeros_like(A)
    outZ = torch.zeros((n, oh, ow), dtype=torch.int32, device=dev)
    outT = torch.zeros((n, oh, ow), dtype=torch.int32, device=dev)
    buf = bytearray(264)

    def put(off, fmt, *vals):
        import struct
        struct.pack_into(fmt, buf, off, *vals)
    ptr = lambda t: 0 if t is None else t.data_ptr()
    Ht, Wt = (int(h.tex.shape[-3]), int(h.tex.shape[-2])) if h.tex is not None else (0, 0)
    put(0, "<7Q4i", ptr(h.pos), ptr(h.vnormals), ptr(h.faces), ptr(h.uv), ptr(h.uv_idx), ptr(h.tex), 0 if h.tex is not None else ptr(h.vertex_color),
        V, Tn, Ht, Wt)
    put(72, "<2Q", P.data_ptr(), bb.data_ptr())
    put(88, "<9f", *[float(v) for v in np.asarray(K, dtype=np.float64).reshape(9).astype(np.float32)])
    inv_r = float(np.float32(1.0) / (np.float32(diam) * np.float32(0.5)))
    put(124, "<5i4fi", H, W, oh, ow, nstrips, 0.8, 0.5, inv_r, 0.001, 3)
    put(168, "<7Q", outA.data_ptr(), 0, 0, 0, 0, outZ.data_ptr(), outT.data_ptr())
    base = ws.data_ptr()
    put(224, "<5Q", base + o_vr, base + o_va, base + o_cnt, base + o_lst, 0)
    kbuf = (C.c_char * len(buf)).from_buffer(buf)
    size = C.c_size_t(len(buf))
    extra = (C.c_void_p * 5)(1, C.cast(kbuf, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
    lds = 16 * ow * 8 + 16 + 96 * 56

    x = torch.randn((14800, 512), device=dev, dtype=torch.float16)
    w = torch.randn((512, 512), device=dev, dtype=torch.float16)
    y = torch.empty((14800, 512), device=dev, dtype=torch.float16)
    torch.matmul(x, w.t(), out=y)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    vdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants")
    notes = json.load(open(os.path.join(vdir, "variants.json")))["variants"]
    for path in sorted(glob.glob(os.path.join(vdir, "*.hsaco"))):
        name = os.path.basename(path)[:-6]
        img = open(path, "rb").read()
        mod, fn = C.c_void_p(), C.c_void_p()
        ck(hip.hipModuleLoadData(C.byref(mod), img), "hipModuleLoadData")
        ck(hip.hipModuleGetFunction(C.byref(fn), mod, KERNEL), "hipModuleGetFunction")

        def launch():
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            ck(hip.hipModuleLaunchKernel(fn, nstrips, n, 1, 256, 1, 1, lds, st, None, C.cast(extra, C.c_void_p)), "hipModuleLaunchKernel")
        for t in (outA, outZ, outT):
            t.zero_()
        launch()
        torch.cuda.synchronize()
        alone_ok = bool(torch.equal(outA, ref["A"]) and torch.equal(outZ, ref["zbuf"]) and torch.equal(outT, ref["tri_id"]))
        bad, first = 0, None
        main = torch.cuda.current_stream(dev)
        for rep in range(launches):
            side.wait_stream(main)
            launch()                                   # the rasteriser first, the GEMM arrives while it runs
            with torch.cuda.stream(side):
                torch.matmul(x, w.t(), out=y)
            main.wait_stream(side)
            torch.cuda.synchronize()
            if not (torch.equal(outA, ref["A"]) and torch.equal(outZ, ref["zbuf"]) and torch.equal(outT, ref["tri_id"])):
                bad += 1
                if first is None:
                    d = (outA != ref["A"]).any(dim=1)                      # (n, oh, ow)
                    idx = d.nonzero()
                    px = (idx[:, 1] * ow + idx[:, 2])
                    lanes = sorted(set(int(v) for v in (px % 64).tolist()))      # lane of the resolve loop = pixel index mod 64 (256 threads, 160-wide rows)
                    first = dict(launch=rep, pixels=int(d.sum()), hypotheses=sorted(set(int(v) for v in idx[:, 0].tolist()))[:8], lanes=lanes,
                                 zbuf_equal=bool(torch.equal(outZ, ref["zbuf"])), tri_id_equal=bool(torch.equal(outT, ref["tri_id"])))
        print(json.dumps(dict(variant=name, packed_left=notes.get(name, {}).get("packed_left"), alone_equal=alone_ok, overlapped_launches=launches,
                              differing=bad, first=first, note=notes.get(name, {}).get("note", ""))), flush=True)
        hip.hipModuleUnload(mod)


if __name__ == "__main__":
    main()
