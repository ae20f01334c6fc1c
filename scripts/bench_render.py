"""fp_render_crops alone: us per launch at N = 1 / 16 / 126 / 252 (the library is chosen with FP_AMD_LIB), and a digest of every output so
that two builds can be compared bit for bit:   FP_AMD_LIB=.../libfp_amd_r1.so python scripts/bench_render.py"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from foundationpose_amd import ops

dev = torch.device("cuda:0")
sc = bench.build_scene(dev, 0, 252)
poses = torch.as_tensor(sc["poses"], device=dev)
h = sc["gm"]["_handle"]
for n in (1, 16, 126, 252):
    P = poses[:n].contiguous()
    _, bb = ops.crop_windows(P, sc["K"], sc["diameter"], 1.2, (160, 160))
    A = torch.zeros((n, 6, 160, 160), dtype=torch.float16, device=dev)
    ws = torch.empty(max(16, ops.workspace_bytes(n, h.V, h.T, 160, 160)), dtype=torch.uint8, device=dev)
    run = lambda want=("A",): ops.render_crops(h, P, bb, sc["K"], 480, 640, out_hw=(160, 160), mesh_diameter=sc["diameter"], xyz_thr=0.001,
                                               normalize_xyz=True, A_out=A, workspace=ws, want=want)
    o = run(("A", "zbuf", "tri_id"))
    dig = hashlib.sha1(b"".join(o[k].cpu().numpy().tobytes() for k in ("A", "zbuf", "tri_id"))).hexdigest()[:16]
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 300
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps(dict(n=n, us_per_launch=round(e0.elapsed_time(e1) * 1e3 / reps, 2), outputs_sha1=dig)), flush=True)
