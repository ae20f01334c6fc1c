"""A/B of the rasteriser's resolve kernel (k_raster) over (FP_ENT, FP_PIX, FP_RASTER_THREADS):
    git apply scripts/experiments/r03_raster_mlp_variants.patch     # the variants are not in the product source
    make -C foundationpose_amd/csrc raster_ab && python scripts/raster_variants.py
runs this file once per library (FP_AMD_LIB) and prints, per variant, the time of fp_render_crops at the sub-batch (126) and
batch (252) sizes of the bench and a digest of every output (A fp16, zbuf, tri_id, colour, xyz): all variants must agree bit
for bit with the product library."""
import ctypes as C
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    import torch
    import bench
    from foundationpose_amd import ops, _lib
    dev = torch.device("cuda:0")
    res = {"lib": os.path.basename(os.environ.get("FP_AMD_LIB", "libfp_amd.so"))}
    L = _lib.lib()
    for N in (126, 252):
        sc = bench.build_scene(dev, 0, N)
        h = sc["gm"]["_handle"]
        poses = torch.as_tensor(sc["poses"], device=dev)
        # spread the hypotheses a little so that the strips differ between hypotheses as they do after an update
        poses[:, :3, 3] += torch.linspace(-0.01, 0.01, N, device=dev)[:, None]
        tf, bb = ops.crop_windows(poses, sc["K"], sc["diameter"], 1.2, (160, 160))
        A = torch.empty((N, 6, 160, 160), dtype=torch.float16, device=dev)
        col = torch.empty((N, 160, 160, 3), dtype=torch.float32, device=dev)
        xyz = torch.empty_like(col)
        zb = torch.empty((N, 160, 160), dtype=torch.int32, device=dev)
        ti = torch.empty_like(zb)
        ws = torch.empty(L.fp_workspace_bytes(N, h.V, h.T, 160, 160), dtype=torch.uint8, device=dev)
        K9 = np.ascontiguousarray(np.asarray(sc["K"], np.float64).reshape(9).astype(np.float32))

        def run(full):
            st = L.fp_render_crops(h.handle, poses.data_ptr(), bb.data_ptr(), K9.ctypes.data_as(C.c_void_p), 480, 640, N, 160, 160,
                                   0.8, 0.5, float(sc["diameter"]), 0.001, 3, A.data_ptr(), col.data_ptr() if full else None, None,
                                   xyz.data_ptr() if full else None, None, zb.data_ptr() if full else None,
                                   ti.data_ptr() if full else None, ws.data_ptr(), ws.numel(),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert st == 0, L.fp_last_error()
        run(True)
        torch.cuda.synchronize()
        dg = hashlib.sha1()
        for t in (A, zb, ti, col, xyz):
            dg.update(t.cpu().numpy().tobytes())
        res[f"digest_{N}"] = dg.hexdigest()[:16]
        res[f"covered_{N}"] = float((ti >= 0).float().mean())
        for _ in range(5):
            run(False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                run(False)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 40 * 1e3)
        res[f"us_{N}"] = round(sorted(ts)[len(ts) // 2], 2)
        res[f"us_min_{N}"] = round(min(ts), 2)
    print("RV " + json.dumps(res), flush=True)


if __name__ == "__main__":
    if os.environ.get("FP_RV_CHILD"):
        one()
        sys.exit(0)
    csrc = os.path.join(ROOT, "foundationpose_amd", "csrc")
    libs = [os.path.join(csrc, "libfp_amd.so")] + sorted(glob.glob(os.path.join(csrc, "libfp_amd_rv_*.so")))
    rows = []
    for lib in libs:
        env = dict(os.environ, FP_AMD_LIB=lib, FP_RV_CHILD="1")
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("RV ")]
        if not line:
            print(f"{os.path.basename(lib)}: FAILED\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}", flush=True)
            continue
        rows.append(json.loads(line[0][3:]))
        print(line[0], flush=True)
    ref = rows[0] if rows else None
    for r in rows[1:]:
        same = all(r[k] == ref[k] for k in ("digest_126", "digest_252"))
        print(f"{r['lib']}: outputs {'identical to' if same else 'DIFFER from'} {ref['lib']}", flush=True)
