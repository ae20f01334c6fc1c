#!/usr/bin/env python
"""Headline benchmark: pose-hypotheses/sec (raster+refine+score), 252 hyp x 160x160 @ 640x480 RGB-D.

A step = one pass of the hot path for one object and one frame: R=5 refine iterations (crop windows -> fused
rasteriser -> observed crop -> RefineNet -> pose update) + one score pass (-> ScoreNet -> ranking), inputs resident
in HBM (BASELINE.json configs[1]).

Multi-GPU (one process per GPU, RCCL):
  --mode object (default; BASELINE configs[3]): every rank owns one object and its 252 hypotheses; the per-object
      {scores, refined poses} records are exchanged with ONE all-gather per step; per-GPU work is fixed (weak scaling).
  --mode hypothesis: the 252 hypotheses of ONE object are sharded over the ranks (dist.register_hypothesis_parallel):
      refinement is embarrassingly parallel, the scorer's cross-hypothesis attention is the one exchange step
      (ONE all-gather of [feature | pose]); total work is fixed (strong scaling).
`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks on
127.0.0.1 and fails loudly when the node has fewer than N GPUs; under a launcher (WORLD_SIZE set) --gpus must equal
the world size.  Rank 0 prints one JSON line.

The headline (`value`, `ms_per_step`) is timed over --steps steps with nothing but the product on the stream; the
per-kernel table (`kernels`, `roofline`) comes from a SECOND pass of the same length with two HIP events around every
entry point (ops.KernelTimers), outside the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA
MFMA_ONLY_UNDER_CAP_TFLOPS = 1800.0   # measured: MFMA-only loop, lane-dependent fp16 operands, 1.4 kW cap (profiles/r06_i_conv_loop_probe.log)
REFINE_GFLOP_PER_HYP = 23.946
SCORE_GFLOP_PER_HYP = 21.938
SCORE_GFLOP_CROSS_252 = 0.659
STEM_GFLOP_PER_IMAGE = (80 * 80 * 64 * 6 * 49 + 40 * 40 * 128 * 64 * 9 + 4 * 40 * 40 * 128 * 128 * 9) * 2 / 1e9   # encodeA, one 160x160 image
# roofline that bounds each hand-written kernel (DESIGN.md "Kernels")
KERNEL_BOUND = {"fp_render_crops": "hbm", "fp_warp_crops": "hbm", "fp_conv7x7s2_bn_relu_fwd": "hbm",
                "fp_igemm_f16_fwd": "mfma", "fp_linear512_f16_fwd": "mfma", "fp_linear_layernorm_fwd": "mfma", "fp_ffn_layernorm_mean_fwd": "mfma", "fp_encoder_tail_mean_fwd": "mfma", "fp_layernorm_res_fwd": "hbm", "fp_add_pe_f16_fwd": "hbm",
                "fp_colmean_f16_fwd": "hbm", "fp_attention_f16_fwd": "mfma"}


def measured_traffic(kernel):
    """-> (HBM bytes per launch, launches it averages, {rocprof kernel name: launches}) from the committed rocprofv3 PMC passes
    over one step of `bench.py --serialize` (profiles/traffic.json, written by scripts/pmc_step_traffic.py with the gfx950
    FETCH_SIZE x2 correction): the mean over THE launches `avg_launch_ms` averages; (None, 0, {}) when not profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, 0, {}
    with open(path) as f:
        ent = json.load(f).get(kernel, {})
    return ent.get("hbm_bytes_per_launch"), ent.get("launches", 0), {k: v.get("launches") for k, v in ent.get("kernels", {}).items()}


def build_scene(dev, seed, n_hyp):
    """one synthetic object + frame on the device; frame rendered with the product's own rasteriser"""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.Utils import euler_matrix, make_mesh_tensors, nvdiffrast_render, sample_views_icosphere
    from foundationpose_amd.mesh import make_can_mesh
    mesh = make_can_mesh(seed=seed)
    gm = make_mesh_tensors(mesh, device=dev)
    K = syn.YCBV_K.copy()
    T = syn.gt_pose(seed)
    color, depth, _ = nvdiffrast_render(K=K, H=syn.H, W=syn.W, ob_in_cams=torch.as_tensor(T[None], device=dev, dtype=torch.float),
                                        mesh_tensors=gm, use_light=True, extra={})
    rgb, d, mask = syn.compose_frame(color[0].cpu().numpy(), depth[0].cpu().numpy(), seed=seed)
    cams = sample_views_icosphere(40)
    grid = np.asarray([np.linalg.inv(c @ euler_matrix(0, 0, a)) for c in cams for a in np.deg2rad(np.arange(0, 360, 60))])
    grid[:, :3, 3] = T[:3, 3] + np.array([0.004, -0.003, 0.01])
    reps = int(np.ceil(n_hyp / len(grid)))
    grid = np.tile(grid, (reps, 1, 1))[:n_hyp]
    diameter = float(np.linalg.norm(mesh.vertices.max(0) - mesh.vertices.min(0)))
    return dict(mesh=mesh, gm=gm, K=K, rgb=rgb, depth=d, mask=mask, poses=grid.astype(np.float32), diameter=diameter, T=T)


def _log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(n_hyp=16, iters=5, reps=3):
    """BASELINE configs[0] (16 hypotheses, one frame, 5 refine iterations + 1 score pass, no GPU) on the oracle -- C
    rasteriser/warp with OpenMP + torch-CPU fp32 networks -- repeated `reps` times; checker, not product"""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.Utils import euler_matrix, sample_views_icosphere
    from foundationpose_amd.mesh import make_can_mesh
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict
    from oracle import ops as oo
    from oracle import pipeline as op
    mesh = make_can_mesh()
    mnp = op.mesh_tensors_np(mesh)
    K, T = syn.YCBV_K, syn.gt_pose(0)
    full = oo.render_crops(mnp, T[None].astype(np.float32), None, K, syn.H, syn.W, (syn.H, syn.W), normalize_xyz=False,
                           want=("color", "depth"))
    rgb, depth, _ = syn.compose_frame(full["color"][0], full["depth"][0])
    cams = sample_views_icosphere(40)
    grid = np.asarray([np.linalg.inv(c @ euler_matrix(0, 0, a)) for c in cams for a in np.deg2rad(np.arange(0, 360, 60))])
    grid[:, :3, 3] = T[:3, 3] + np.array([0.004, -0.003, 0.01])
    poses = grid[:n_hyp].astype(np.float32)
    diam = float(np.linalg.norm(mesh.vertices.max(0) - mesh.vertices.min(0)))
    rcfg, scfg = dict(DEFAULT_REFINE_CFG), dict(DEFAULT_SCORE_CFG)
    rsd, ssd = random_state_dict("refine", rcfg, 0), random_state_dict("score", scfg, 0)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32)
    torch.set_num_threads(cores)
    oo.set_num_threads(cores)
    d = op.preprocess_depth(depth)
    xyz = oo.depth2xyzmap(d, K)
    op.refine_predict(rcfg, rsd, rgb, d, K, poses[:2], xyz, mnp, diam, iteration=1)  # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        p = op.refine_predict(rcfg, rsd, rgb, d, K, poses, xyz, mnp, diam, iteration=iters)
        s = op.score_predict(scfg, ssd, rgb, d, K, p, mnp, diam)
        np.argsort(-s)
    dt = time.perf_counter() - t0
    return dict(value=reps * n_hyp / dt, unit="pose-hypotheses/sec", cores=int(cores), kind="port",
                sample=f"BASELINE configs[0] x {reps}: {n_hyp} hypotheses x ({iters} refine + 1 score), one frame, oracle C "
                       f"raster/warp (OpenMP {oo.num_threads()} thr) + torch-CPU fp32 nets ({torch.get_num_threads()} thr), {dt:.2f} s")


def ingest_bench(dev, sc, reps=200):
    """per-frame ingest (erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch, estimater.py:256-258), the
    three launches back to back on one stream, HIP events around `reps` frames; input resident in HBM"""
    from foundationpose_amd import ops
    d = torch.as_tensor(sc["depth"], device=dev)

    def pre():
        f = ops.bilateral_filter_depth(ops.erode_depth(d, radius=2), radius=2)
        return ops.depth_to_xyz(f, sc["K"], zfar=float("inf"), f64_internal=False)
    for _ in range(5):
        pre()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pre()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    H, W = d.shape
    algo = H * W * 4 * (2 + 2 + 1 + 3)        # read + write per filter, read depth + write xyz
    return {"us_per_frame": us, "frames_per_sec": 1e6 / us, "algorithmic_bytes": algo, "GBps": algo / us / 1e3,
            "frac_of_hbm_peak": algo / us / 1e3 / HBM_PEAK_GBS, "frame": [int(H), int(W)],
            "note": "erode (r=2) + bilateral (r=2) + back-projection, 3 launches, launch-latency bound at this size (1.2 MB frame)"}


def make_sequence(dev, sc, frames, seed=7):
    """config 5's input: `frames` DISTINCT RGB-D frames of the object on a smooth trajectory (<= 4 mm, <= 1.5 deg per frame), each with
    its own noise / dropout, rendered by the product's rasteriser; pinned host buffers (the H2D copies belong to the timed region)"""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.Utils import nvdiffrast_render
    rng = np.random.default_rng(seed)
    T0 = sc["T"].astype(np.float64)
    gt = np.zeros((frames, 4, 4))
    for f in range(frames):
        a = 2 * np.pi * f / 250.0
        ax = np.array([np.sin(0.7 * a), np.cos(a), 0.3])
        ax /= np.linalg.norm(ax)
        ang = np.deg2rad(20.0) * np.sin(a)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        gt[f] = T0
        gt[f, :3, :3] = dR @ T0[:3, :3]
        gt[f, :3, 3] = T0[:3, 3] + np.array([0.06 * np.sin(a), 0.04 * np.sin(2 * a), 0.05 * np.cos(a) - 0.05])
    rgb_h = torch.empty((frames, syn.H, syn.W, 3), dtype=torch.uint8).pin_memory()
    depth_h = torch.empty((frames, syn.H, syn.W), dtype=torch.float32).pin_memory()
    gen = torch.Generator(device=dev).manual_seed(11)
    bg = torch.as_tensor(np.kron(rng.uniform(0.3, 0.6, size=(syn.H // 8, syn.W // 8, 3)), np.ones((8, 8, 1))), device=dev, dtype=torch.float32)
    t0 = time.perf_counter()
    for f in range(frames):
        color, depth, _ = nvdiffrast_render(K=sc["K"], H=syn.H, W=syn.W, ob_in_cams=torch.as_tensor(gt[f][None], device=dev, dtype=torch.float),
                                            mesh_tensors=sc["gm"], use_light=True, extra={})
        mask = depth[0] > 0
        rgb = torch.where(mask[..., None], color[0], bg)
        d = torch.where(mask, depth[0], torch.full_like(depth[0], 1.2)) + torch.randn((syn.H, syn.W), generator=gen, device=dev) * 0.001
        d = torch.where(torch.rand((syn.H, syn.W), generator=gen, device=dev) < 0.02, torch.zeros_like(d), d)
        rgb_h[f].copy_((rgb.clamp(0, 1) * 255).to(torch.uint8))
        depth_h[f].copy_(d)
    torch.cuda.synchronize()
    return gt, rgb_h, depth_h, time.perf_counter() - t0


def tracking_bench(dev, sc, refiner, seq, hyps, iters, latency_frames=200):
    """BASELINE configs[4] for one (hypotheses per frame, iterations) pair: per frame the uint8 colour image, the
    float depth map and the hypotheses are uploaded from pinned host memory, then depth erosion + bilateral filter + back-projection +
    `iters` x (crop windows, rasteriser, observed crop, RefineNet, pose update) run as captured hipGraphs (graphs.GraphedTracker) or
    as the same launches issued eagerly.  hyps = 1, iters = 2 is the reference's track_one (estimater.py:250-268, run_demo.py:64)."""
    from foundationpose_amd import synthetic as syn
    from foundationpose_amd.graphs import GraphedTracker
    gt, rgb_h, depth_h, t_gen = seq
    F_ = len(gt)
    hyp_h = torch.empty((F_, hyps, 4, 4), dtype=torch.float32).pin_memory()
    for f in range(F_):
        if hyps == 1:
            hyp_h[f, 0].copy_(torch.from_numpy(gt[max(f - 1, 0)].astype(np.float32)))      # track_one: the previous frame's pose
        else:                                                                              # <= 2 cm, <= 10 deg around it
            hyp_h[f].copy_(torch.from_numpy(syn.perturbed_poses(gt[max(f - 1, 0)], hyps, seed=100 + f, max_trans=0.02, max_rot_deg=10.0).astype(np.float32)))
    trk = GraphedTracker(refiner, sc["gm"], sc["diameter"], sc["K"], syn.H, syn.W, n_hyp=hyps, iteration=iters, device=dev).capture()
    rgb_u8 = torch.empty((syn.H, syn.W, 3), dtype=torch.uint8, device=dev)

    def frame(f, graph=True):
        rgb_u8.copy_(rgb_h[f], non_blocking=True)            # H2D, 0.92 MB
        trk.rgb.copy_(rgb_u8)                                # u8 -> f32 on the device
        trk.depth.copy_(depth_h[f], non_blocking=True)       # H2D, 1.2 MB
        trk.poses_in.copy_(hyp_h[f], non_blocking=True)      # H2D
        return trk.replay() if graph else trk._body()
    res, lat = {}, {}
    with torch.inference_mode():
        for name, graph in (("hipgraph", True), ("eager", False)):
            for f in range(5):
                frame(f, graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(F_):
                out = frame(f, graph)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / F_
            # per-frame latency: the host waits for every frame, as a control loop would
            ls = []
            for f in range(min(F_, latency_frames)):
                t0 = time.perf_counter()
                frame(f, graph)
                torch.cuda.synchronize()
                ls.append(time.perf_counter() - t0)
            lat[name] = {"median": float(np.median(ls) * 1e3), "p95": float(np.percentile(ls, 95) * 1e3)}
        assert torch.isfinite(out).all()
        # round 6: the same loop as a two-slot pipeline (graphs.FramePipeline): upload + u8 -> f32 + erode + bilateral + back-projection
        # of frame f + 1 on an ingest stream under the refine loop of frame f; frame f + 1's refine still waits for frame f's
        from foundationpose_amd.graphs import FramePipeline
        pipe = FramePipeline(trk)

        def run_pipelined(n, sync_each=False, keep=0, graph=True):
            outs, ls = [], []
            pipe.submit(0, rgb_h[0], depth_h[0], hyp_h[0])
            for f in range(n):
                t1 = time.perf_counter()
                if f + 1 < n:
                    pipe.submit((f + 1) % 2, rgb_h[f + 1], depth_h[f + 1], hyp_h[f + 1])
                o = pipe.run(f % 2, graph=graph)
                if f < keep:
                    outs.append(o.clone())
                if sync_each:
                    torch.cuda.synchronize()
                    ls.append(time.perf_counter() - t1)
            torch.cuda.synchronize()
            return outs, ls
        run_pipelined(5)
        t0 = time.perf_counter()
        run_pipelined(F_)
        res["pipelined"] = (time.perf_counter() - t0) / F_
        run_pipelined(5, graph=False)
        t0 = time.perf_counter()
        run_pipelined(F_, graph=False)
        res["pipelined_eager"] = (time.perf_counter() - t0) / F_
        _, ls = run_pipelined(min(F_, latency_frames), sync_each=True)
        lat["pipelined"] = {"median": float(np.median(ls) * 1e3), "p95": float(np.percentile(ls, 95) * 1e3)}
        keep = min(F_, 12)
        po, _ = run_pipelined(keep, keep=keep)
        same = all(bool(torch.equal(po[f], frame(f, True))) for f in range(keep))
    return {"frames": F_, "hypotheses_per_frame": hyps, "refine_iterations": iters, "distinct_frames": F_,
            "pipelined_ms_per_frame": res["pipelined"] * 1e3, "frames_per_sec_pipelined": 1.0 / res["pipelined"],
            "pipelined_eager_ms_per_frame": res["pipelined_eager"] * 1e3,
            "latency_ms_synced_per_frame_pipelined": lat["pipelined"], "pipelined_poses_equal_unpipelined": bool(same),
            "ingest_stream_overlaps": pipe.ingest_overlaps,
            "uploads_per_frame_bytes": int(rgb_h[0].numel() + depth_h[0].numel() * 4 + hyp_h[0].numel() * 4),
            "hipgraph_ms_per_frame": res["hipgraph"] * 1e3, "eager_ms_per_frame": res["eager"] * 1e3,
            "frames_per_sec": 1.0 / res["hipgraph"], "frames_per_sec_eager": 1.0 / res["eager"],
            "hypothesis_passes_per_sec": hyps * iters / res["hipgraph"], "speedup_vs_eager": res["eager"] / res["hipgraph"],
            "latency_ms_synced_per_frame": lat["hipgraph"], "latency_ms_synced_per_frame_eager": lat["eager"]}


def _respawn(args):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run"""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} requested but this node exposes {n_dev} GPU(s); refusing to run fewer ranks")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


class ClockSampler:
    """samples the GPU's shader clock (and socket power) from the amdgpu hwmon files while the timed region runs, so that a
    box-to-box difference of the step time can be attributed (the chip clocks to its power budget: MI355X_MICROARCH.md,
    "DVFS give-back").  Nothing here touches the GPU; a missing file gives nulls."""

    def __init__(self, index=0, period_s=0.005):
        import glob
        import threading
        self.freq = self.power = None
        cards = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq*_input")):
            lab = self._read_str(f.replace("_input", "_label"))
            if lab == "sclk" and self._read_str(os.path.join(os.path.dirname(f), "name")) == "amdgpu":
                cards.append(f)            # one per GPU: the shader clock of an amdgpu device
        # the node's sysfs lists every GPU of the host, the process may see one of them: pick the card whose PCI address is
        # the torch device's (falls back to the index-th card)
        pick = None
        try:
            pr = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for f in cards:
                if addr in os.path.realpath(f):
                    pick = f
        except Exception:
            pick = None
        self.matched_pci = pick is not None
        if cards:
            self.freq = pick or cards[min(index, len(cards) - 1)]
            d = os.path.dirname(self.freq)
            pw = [f for f in (os.path.join(d, "power1_average"), os.path.join(d, "power1_input")) if os.path.exists(f)]
            self.power = pw[0] if pw else None
            # one-shot readings next to the sampled shader clock: memory clock, junction temperature, power cap
            self.static = {}
            for f in sorted(glob.glob(os.path.join(d, "freq*_input"))):
                lab = self._read_str(f.replace("_input", "_label"))
                if lab and lab != "sclk":
                    v = self._read(f)
                    self.static[lab + "_MHz"] = None if v is None else v / 1e6
            for name, f, scale in (("temp_junction_C", "temp2_input", 1e3), ("power_cap_W", "power1_cap", 1e6)):
                v = self._read(os.path.join(d, f))
                if v is not None:
                    self.static[name] = v / scale
        self.period, self.f, self.p = period_s, [], []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read_str(path):
        try:
            with open(path) as fh:
                return fh.read().strip()
        except Exception:
            return None

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read(self.freq) if self.freq else None
            if v is not None:
                self.f.append(v / 1e6)
            w = self._read(self.power) if self.power else None
            if w is not None:
                self.p.append(w / 1e6)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.freq:
            self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.freq:
            self._thread.join(timeout=1.0)

    def summary(self):
        if not self.f:
            return {"sclk_MHz_mean": None, "samples": 0, "source": self.freq}
        import statistics
        out = {"sclk_MHz_mean": statistics.fmean(self.f), "sclk_MHz_min": min(self.f), "sclk_MHz_max": max(self.f),
               "samples": len(self.f), "source": self.freq, "source_matches_torch_device_pci": bool(getattr(self, "matched_pci", False))}
        if self.p:
            out["power_W_mean"] = statistics.fmean(self.p)
        out.update(getattr(self, "static", {}))
        return out


def _flush_c_stdio():
    """RCCL announces itself through C stdio ("Librccl path : ..."), which a pipe buffers until exit: every rank empties that
    buffer before the final barrier, so that rank 0's JSON line is the last line of the job's stdout"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


_AB_OVERRIDES = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hyps", type=int, default=252)
    ap.add_argument("--refine-iters", type=int, default=5)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32", "torch_amp"],
                    help="fp16: the deployed plan, every network op on libfp_amd.so; torch_amp: the nn.Module under torch.autocast on "
                         "PyTorch-ROCm (MIOpen / rocBLAS) behind the same predictors = BASELINE configs[1] literally; fp32: parity config")
    ap.add_argument("--mode", default="object", choices=["object", "hypothesis"])
    ap.add_argument("--streams", type=int, default=2, help="hypothesis sub-batches run on concurrent HIP streams (1: none)")
    ap.add_argument("--serialize", action="store_true", help="issue the sub-batches on ONE stream in the timed region too "
                    "(the launches of the per-kernel table; used for the rocprofv3 profile that table is checked against)")
    ap.add_argument("--trace-markers", action="store_true", help="bracket the timed region with two marker launches (k_depth_to_xyz on a "
                    "1 x 7 image) so that scripts/concurrent_roofline.py can cut it out of a rocprofv3 --kernel-trace of this command")
    ap.add_argument("--shared-crop", action="store_true", help="tell the refiner that all hypotheses start at one translation, as "
                    "estimater.register() does: the observed crop of the first iteration is then warped and stem-encoded once per "
                    "sub-batch instead of once per hypothesis (bit-identical result).  Off by default: the headline runs every "
                    "hypothesis's full arithmetic")
    ap.add_argument("--no-graph", action="store_true", help="eager launches in the timed region instead of hipGraph replays of the "
                    "refine loop and of the scorer's per-hypothesis half (PoseRefinePredictor / ScorePredictor graph=False)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the records next to the headline: per-frame ingest (a1-a3) and tracking "
                    "mode (BASELINE configs[4] at 64 hypotheses per frame, and the reference's track_one: 1 hypothesis x 2 iterations)")
    ap.add_argument("--track-frames", type=int, default=1000)
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the second (instrumented) pass")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _respawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus})")
    import torch.distributed as dist
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs cuda:{local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    use_dist = world > 1 or os.environ.get("FP_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL on one GPU
    if args.streams > 1:
        from foundationpose_amd.overlap import reserve_streams
        reserve_streams(dev, args.streams - 1)      # before RCCL creates its streams (overlap.reserve_streams)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus

    from foundationpose_amd import ops
    from foundationpose_amd.predict_pose_refine import PoseRefinePredictor
    from foundationpose_amd.predict_score import ScorePredictor
    from foundationpose_amd.weights import DEFAULT_REFINE_CFG, DEFAULT_SCORE_CFG, random_state_dict

    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
    # A/B runs of the measuring scripts (scripts/gpu.sh): FP_BENCH_ENGINE="PACKED_CONV_TILES=0,FUSED_FFN=0" flips switches of engine.py
    # for this process through engine.overrides -- the package itself reads no environment variable
    ab = os.environ.get("FP_BENCH_ENGINE", "").strip()
    if ab:
        from foundationpose_amd import engine
        _AB_OVERRIDES.append(engine.overrides(**{k.strip(): int(v) for k, v in (kv.split("=") for kv in ab.split(","))}))
        _AB_OVERRIDES[-1].__enter__()          # kept alive for the life of the process: a collected generator would restore the switches
    N, R = args.hyps, args.refine_iters
    hyp_mode = args.mode == "hypothesis"
    _log("building scene")
    sc = build_scene(dev, seed=0 if hyp_mode else rank, n_hyp=N)   # hypothesis mode: every rank sees the same object
    # graph=True: captured at the first (warm-up) call of a key; the passes under ops.KernelTimers launch eagerly by themselves
    opts = dict(device=dev, precision=args.precision, n_streams=args.streams, graph=not args.no_graph)
    refiner = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), **opts)
    scorer = ScorePredictor(cfg=dict(DEFAULT_SCORE_CFG), state_dict=random_state_dict("score", seed=0), **opts)
    rgb_t = torch.as_tensor(sc["rgb"], device=dev).float().contiguous()
    depth_t = ops.bilateral_filter_depth(ops.erode_depth(torch.as_tensor(sc["depth"], device=dev)))
    xyz_t = ops.depth_to_xyz(depth_t, sc["K"], f64_internal=True)
    poses0 = torch.as_tensor(sc["poses"], device=dev)
    from foundationpose_amd.dist import gather_object_records, register_hypothesis_parallel

    def step():
        if hyp_mode:
            p, s, _ = register_hypothesis_parallel(refiner, scorer, rgb_t, depth_t, sc["K"], poses0, xyz_t, mesh=sc["mesh"],
                                                   mesh_tensors=sc["gm"], mesh_diameter=sc["diameter"], iteration=R,
                                                   shared_translation=args.shared_crop)
            return torch.cat([s.reshape(-1, 1), p.reshape(-1, 16)], dim=1)[None]   # replicated on every rank
        p, _ = refiner.predict(rgb_t, depth_t, sc["K"], poses0, xyz_t, mesh=sc["mesh"], mesh_tensors=sc["gm"],
                               mesh_diameter=sc["diameter"], iteration=R, shared_translation=args.shared_crop)
        s, _ = scorer.predict(rgb_t, depth_t, sc["K"], p, mesh=sc["mesh"], mesh_tensors=sc["gm"],
                              mesh_diameter=sc["diameter"])
        ids = s.argsort(descending=True)
        return gather_object_records(s[ids], p[ids])  # ONE RCCL all-gather of [score|pose] per register()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.serialize:
        refiner.sub.serial = scorer.sub.serial = True
    _log("warmup")
    for _ in range(args.warmup):
        step()
    sync()
    _log("timed region")
    clock = ClockSampler(local_rank)
    marker_in = torch.ones((1, 7), dtype=torch.float32, device=dev)

    def marker():       # outside the clock: the timed region below is exactly the K steps either way
        if args.trace_markers:
            ops.depth_to_xyz(marker_in, sc["K"])
            torch.cuda.synchronize()
    marker()
    with clock:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rec = step()
        sync()
        dt = time.perf_counter() - t0
    marker()
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert torch.isfinite(rec).all()
    # not the headline: the same step the way estimater.register() issues it -- every hypothesis of the rotation grid starts at ONE
    # translation (estimater.py:132-133), so the observed crop of the first refine iteration is warped and stem-encoded once per
    # sub-batch (bit-identical poses; PoseRefinePredictor.predict(shared_translation=True)).  Reported next to the headline so that a
    # driver-run record shows it; skipped when the headline already runs that way (--shared-crop) or in hypothesis mode.
    dt_reg = None
    if not args.shared_crop and not hyp_mode and args.precision == "fp16" and R > 0:
        def step_register():
            p, _ = refiner.predict(rgb_t, depth_t, sc["K"], poses0, xyz_t, mesh=sc["mesh"], mesh_tensors=sc["gm"],
                                   mesh_diameter=sc["diameter"], iteration=R, shared_translation=True)
            s, _ = scorer.predict(rgb_t, depth_t, sc["K"], p, mesh=sc["mesh"], mesh_tensors=sc["gm"], mesh_diameter=sc["diameter"])
            ids = s.argsort(descending=True)
            return gather_object_records(s[ids], p[ids])
        for _ in range(2):
            step_register()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_register()
        sync()
        dt_reg = time.perf_counter() - t0
    # second, instrumented pass (not part of the headline): per-entry-point HIP events on the launch stream
    timers = ops.KernelTimers()
    if rank == 0 and not args.no_kernel_table:
        _log("instrumented pass")
    if not args.no_kernel_table:
        # the same launches (same sub-batches, same sizes), but issued on ONE stream: a HIP-event pair around a kernel that
        # shares the chip with another stream's kernels measures the mix, not the kernel
        refiner.sub.serial = scorer.sub.serial = True
        clock_instr = ClockSampler(local_rank)
        with clock_instr:
            with timers:
                for _ in range(args.steps):
                    step()
            sync()
        refiner.sub.serial = scorer.sub.serial = args.serialize
    # third pass, only when the step is split: the same kernels in ONE launch sequence over all hypotheses -- the launch
    # sizes of `--streams 1` and of rounds 1 / 2a, for a like-for-like per-launch figure of the dominant kernel
    timers_full = ops.KernelTimers()
    # whether the step is split is decided per process by a stream-overlap probe + an exactness canary (overlap.py); every
    # extra pass below contains the step's collective, so under torch.distributed all ranks must take the same branch
    split = torch.tensor([1 if len(refiner.sub.parts(N, dev)) > 1 else 0], device=dev, dtype=torch.int32)
    if use_dist:
        dist.all_reduce(split, op=dist.ReduceOp.MIN)
    split = bool(split.item())
    if not args.no_kernel_table and split:
        ns = refiner.sub.n_streams
        refiner.sub.n_streams = scorer.sub.n_streams = 1
        step()
        with timers_full:
            for _ in range(args.steps):
                step()
        sync()
        refiner.sub.n_streams = scorer.sub.n_streams = ns

    # fourth pass: the launches of the TIMED mode (sub-batches on concurrent streams) with HIP events on their own streams:
    # busy time of the dominant kernel = union of its launch intervals, i.e. its roofline in the execution mode that is timed
    timers_conc, conc_ref, conc_wall = ops.KernelTimers(), None, None
    if not args.no_kernel_table and not args.serialize and split:
        step()
        sync()
        conc_ref = torch.cuda.Event(enable_timing=True)
        conc_end = torch.cuda.Event(enable_timing=True)
        conc_ref.record()
        with timers_conc:
            for _ in range(args.steps):
                step()
        conc_end.record()
        sync()
        conc_wall = conc_ref.elapsed_time(conc_end)

    total_hyps = N if hyp_mode else world * N
    if rank == 0:
        V, T = sc["gm"]["_handle"].V, sc["gm"]["_handle"].T
        flops = total_hyps * (R * REFINE_GFLOP_PER_HYP + SCORE_GFLOP_PER_HYP) + (1 if hyp_mode else world) * SCORE_GFLOP_CROSS_252 * (N / 252.0) ** 2
        if args.shared_crop and args.precision == "fp16" and R > 0:
            # arithmetic NOT executed with --shared-crop: the stem (patch-embed + 5 convs) of all but one observed crop per part
            per_rank = (N + world - 1) // world if hyp_mode else N
            flops -= world * max(0, per_rank - max(1, args.streams)) * STEM_GFLOP_PER_IMAGE
        out = {
            "metric": "pose-hypotheses/sec (raster+refine+score), 252 hyp x 160x160 @ 640x480 RGB-D",
            "value": total_hyps * args.steps / dt, "unit": "pose-hypotheses/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if hyp_mode else "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: synthetic can (V={V}, T={T}), one 640x480 RGB-D frame per rank, "
                                   f"{N} hypotheses, {R} refine iterations + 1 score pass, 160x160 crops, random-init weights",
                       "hypotheses_per_gpu": (N + world - 1) // world if hyp_mode else N, "refine_iterations": R,
                       "shared_observed_crop_in_iteration_0": bool(args.shared_crop and args.precision == "fp16"),
                       "launch_mode": ("eager launches" if (args.no_graph or args.precision != "fp16") else
                                       "hipGraph replay: one linear graph per sub-batch for the 5-iteration refine loop and one for the "
                                       "scorer's per-hypothesis half, bit-identical to the eager launches"),
                       "network": {"fp16": "libfp_amd.so (hand-written MFMA kernels)", "torch_amp": "PyTorch-ROCm under torch.autocast "
                                   "(MIOpen / rocBLAS / ATen)", "fp32": "PyTorch-ROCm fp32"}[args.precision],
                       "parallelism": ("single GPU, no collective (torch.distributed not initialised)" if not use_dist else
                                       f"hypothesis-parallel x{world}: {N} hypotheses of one object sharded, one RCCL all-gather of "
                                       f"[feature|pose] per step" if hyp_mode else
                                       f"object-parallel x{world}, one RCCL all-gather of [score|pose] records per step")},
            "concurrency": {"sub_batches": len(refiner.sub.parts(N, dev)), "rows": [e - a for a, e in refiner.sub.parts(N, dev)],
                            "serialized": bool(args.serialize),
                            "note": "independent hypothesis sub-batches of the step run on concurrent HIP streams in the timed "
                                    "region (foundationpose_amd/overlap.py); the per-kernel table and `roofline` time the same "
                                    "launches issued on one stream"},
            "clock": clock.summary(),
            "energy": None,
            "register_path": None if dt_reg is None else {
                "ms_per_step": dt_reg / args.steps * 1e3, "value": total_hyps * args.steps / dt_reg,
                "note": "the same step as estimater.register() issues it (shared_translation=True: one observed crop per sub-batch in the "
                        "first refine iteration, bit-identical poses); NOT the headline, which runs every hypothesis's full arithmetic"},
            "network_mfma": {"algorithmic_TFLOP_per_step": flops / 1e3, "achieved_TFLOPs": flops / 1e3 / (dt / args.steps),
                             "frac_of_mfma_peak": flops / 1e3 / (dt / args.steps) / (MFMA_PEAK_TFLOPS * world)},
        }
        pw = out["clock"].get("power_W_mean")
        if pw and out["clock"].get("source_matches_torch_device_pci"):
            # the step is power-limited: energy is the number a faster schedule has to move, not only time
            j_step = pw * world * (dt / args.steps)
            out["energy"] = {"socket_power_W_mean": pw, "J_per_step": j_step, "J_per_hypothesis": j_step / total_hyps,
                             "hypotheses_per_joule": total_hyps / j_step, "pJ_per_network_flop": j_step / (flops * 1e9) * 1e12,
                             "note": "amdgpu hwmon socket power sampled every 5 ms through the timed region x step time"}
        if not args.no_kernel_table:
            ksum = timers.summary()
            kern = {}
            for name, k in ksum.items():
                ent = dict(calls=k["calls"], avg_ms=round(k["avg_ms"], 5))
                sec = k["avg_ms"] * 1e-3
                if k["bytes"] > 0 and sec > 0:
                    ent["algorithmic_bytes"] = int(k["bytes"])
                    ent["GBps"] = k["bytes"] / sec / 1e9
                    ent["frac_hbm"] = ent["GBps"] / HBM_PEAK_GBS
                if k["flops"] > 0 and sec > 0:
                    ent["algorithmic_flops"] = k["flops"]
                    ent["TFLOPs"] = k["flops"] / sec / 1e12
                    ent["frac_mfma"] = ent["TFLOPs"] / MFMA_PEAK_TFLOPS
                kern[name] = ent
            # dominant hand-written kernel = largest total time inside the instrumented pass
            dom = max((n for n in ksum if n in KERNEL_BOUND), key=lambda n: ksum[n]["calls"] * ksum[n]["avg_ms"])
            dk, bound = ksum[dom], KERNEL_BOUND[dom]
            sec = dk["avg_ms"] * 1e-3
            if bound == "hbm":
                ach, peak, unit = dk["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
            else:
                ach, peak, unit = dk["flops"] / sec / 1e12, MFMA_PEAK_TFLOPS, "TFLOP/s"
            r_ms, w_ms = ksum["fp_render_crops"]["avg_ms"], ksum["fp_warp_crops"]["avg_ms"]
            stage_bytes = ksum["fp_render_crops"]["bytes"] + ksum["fp_warp_crops"]["bytes"]
            traffic, traffic_n, traffic_kernels = measured_traffic(dom)
            out["roofline"] = {"kernel": dom, "rocprof_kernels": traffic_kernels, "bound": bound, "achieved": ach, "peak": peak,
                               "unit": unit, "frac": ach / peak, "traffic": traffic,
                               "traffic_note": f"HBM-side bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes), mean "
                                               f"over the {traffic_n} launches of this entry point in ONE step of `bench.py --serialize` -- the "
                                               f"launches (kernels, sub-batch sizes, one stream) that achieved / avg_launch_ms average; "
                                               f"algorithmic HBM bytes of the same launches: algorithmic_bytes_per_launch "
                                               f"(scripts/pmc_step_traffic.py -> profiles/traffic.json)",
                               "algorithmic_bytes_per_launch": dk["bytes"],
                               "algorithmic_per_launch": dk["bytes"] if bound == "hbm" else dk["flops"],
                               "avg_launch_ms": dk["avg_ms"], "launches_timed": dk["calls"]}
            if timers_full.records:
                fk = timers_full.summary()[dom]
                fsec = fk["avg_ms"] * 1e-3
                fach = (fk["bytes"] / fsec / 1e9) if bound == "hbm" else (fk["flops"] / fsec / 1e12)
                out["roofline"]["full_batch_launches"] = {
                    "achieved": fach, "frac": fach / peak, "avg_launch_ms": fk["avg_ms"], "launches_timed": fk["calls"],
                    "algorithmic_per_launch": fk["bytes"] if bound == "hbm" else fk["flops"],
                    "note": "the same entry point when the step runs as ONE launch sequence over all hypotheses (--streams 1): "
                            "the launch sizes of the previous rounds' roofline figure"}
            if conc_ref is not None:
                cb = timers_conc.busy(conc_ref)
                ck = cb[dom]
                cach = (ck["bytes"] / (ck["busy_ms"] * 1e-3) / 1e9) if bound == "hbm" else (ck["flops"] / (ck["busy_ms"] * 1e-3) / 1e12)
                out["roofline"]["concurrent"] = {
                    "achieved": cach, "frac": cach / peak, "unit": unit, "busy_ms": ck["busy_ms"], "sum_of_launch_ms": ck["sum_ms"],
                    "launches_timed": ck["calls"], "region_wall_ms": conc_wall, "busy_share_of_region": ck["busy_ms"] / conc_wall,
                    "mean_concurrency": ck["sum_ms"] / max(ck["busy_ms"], 1e-9), "ms_per_step_with_events": conc_wall / args.steps,
                    "note": "the execution mode that is timed: sub-batches on concurrent streams, HIP events recorded on each "
                            "launch's own stream; achieved = algorithmic work of all launches / BUSY time (union of the launch "
                            "intervals).  Cross-check from a rocprofv3 --kernel-trace of the default command: "
                            "scripts/concurrent_roofline.py -> profiles/r03_concurrent_roofline.json",
                    "other_entry_points_busy_ms": {n: round(v["busy_ms"], 3) for n, v in cb.items() if n != dom}}
            ck_ = out["clock"]
            if bound == "mfma" and ck_.get("sclk_MHz_mean") and ck_.get("source_matches_torch_device_pci"):
                # the chip does not hold its 2.4 GHz boost clock under this load (DVFS: MI355X_MICROARCH.md); the MFMA peak scales
                # with the shader clock, so this is the fraction of what the matrix cores could do AT THE CLOCK THEY RAN AT
                pk = peak * ck_["sclk_MHz_mean"] / 2400.0
                out["roofline"]["at_sampled_clock"] = {"sclk_MHz": ck_["sclk_MHz_mean"], "peak": pk, "frac": ach / pk,
                                                       "frac_concurrent": (out["roofline"]["concurrent"]["achieved"] / pk)
                                                       if "concurrent" in out["roofline"] else None,
                                                       "note": "peak x sclk / 2400 MHz; `frac` above stays against the datasheet peak"}
            if bound == "mfma":
                # round 6 (scripts/conv_loop_probe, profiles/r06_i_conv_loop_probe.log): what the 1.4 kW cap leaves an fp16 MFMA kernel on
                # non-trivial operand data on this chip -- a register-resident MFMA-ONLY loop on lane-dependent data sustains 1.80 PFLOP/s
                # (1.72 GHz), the convolution main loop with its LDS / LDS-DMA traffic 1.50-1.55.  A committed measurement, not a peak.
                out["roofline"]["under_power_cap"] = {
                    "mfma_only_sustained": MFMA_ONLY_UNDER_CAP_TFLOPS, "conv_main_loop_sustained": 1550.0, "unit": unit,
                    "frac_of_mfma_only_sustained": ach / MFMA_ONLY_UNDER_CAP_TFLOPS,
                    "source": "profiles/r06_i_conv_loop_probe.log (scripts/conv_loop_probe/probe.hip 2000 random)",
                    "note": "`frac` above stays against the 2.5 PFLOP/s datasheet peak"}
            out["stage_raster_crop"] = {"bytes_per_pass": stage_bytes, "ms_per_pass": r_ms + w_ms,
                                        "achieved_GBps": stage_bytes / ((r_ms + w_ms) * 1e-3) / 1e9,
                                        "frac_of_hbm_peak": stage_bytes / ((r_ms + w_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # at-clock fractions per kernel: the instrumented pass runs one stream at a higher clock than the timed region (less power), so
            # each MFMA-bound entry point is also priced against 2.5 PFLOP/s x (clock sampled during THAT pass) / 2.4 GHz
            ci = clock_instr.summary()
            if ci.get("sclk_MHz_mean") and ci.get("source_matches_torch_device_pci"):
                for name, ent in kern.items():
                    if "TFLOPs" in ent:
                        ent["frac_mfma_at_clock"] = ent["TFLOPs"] / (MFMA_PEAK_TFLOPS * ci["sclk_MHz_mean"] / 2400.0)
                out["clock_instrumented_pass"] = {"sclk_MHz_mean": ci["sclk_MHz_mean"], "power_W_mean": ci.get("power_W_mean")}
            out["kernels"] = kern
            # per-stage rates ("reported separately"): launch time of a stage's entry points per step, one stream, and the
            # hypothesis-passes/s the stage would sustain alone (a pass = one hypothesis through one refine iteration or the score pass)
            groups = {"crop_windows": ("fp_crop_windows",), "raster": ("fp_render_crops",), "observed_crop": ("fp_warp_crops",),
                      "pose_update": ("fp_pose_update",)}
            net = [n for n in ksum if n not in sum(groups.values(), ())]
            groups["networks"] = tuple(net)
            passes = total_hyps // world * (R + 1)
            stages = {}
            for g, names in groups.items():
                ms = sum(ksum[n]["calls"] * ksum[n]["avg_ms"] for n in names if n in ksum) / args.steps
                if ms > 0:
                    stages[g] = {"ms_per_step": ms, "hypothesis_passes_per_sec": passes / (ms * 1e-3)}
            out["stages"] = stages
        if world == 1 and not args.no_extras and args.precision == "fp16":
            # outside the headline's timed region, the same process and clock state: SURVEY 8(d) C5 + the per-frame ingest
            _log("extras: ingest, tracking")
            out["ingest"] = ingest_bench(dev, sc)
            trk_ref = PoseRefinePredictor(cfg=dict(DEFAULT_REFINE_CFG), state_dict=random_state_dict("refine", seed=0), device=dev)
            seq = make_sequence(dev, sc, args.track_frames)
            out["tracking"] = {
                "metric": "tracking frames/sec (BASELINE configs[4]: synthetic RGB-D sequence of distinct frames, per-frame H2D of rgb u8 + "
                          "depth f32 + hypotheses inside the timed region, depth filters + refine loop as hipGraph replays)",
                "sequence_generation_s": seq[3],
                "config5_64hyp_2iter": tracking_bench(dev, sc, trk_ref, seq, 64, 2),
                "track_one_1hyp_2iter": tracking_bench(dev, sc, trk_ref, seq, 1, 2)}
        if world == 1 and not args.no_cpu_baseline:
            _log("cpu baseline (oracle on the host cores)")
            out["cpu_baseline"] = cpu_baseline()
        faulthandler.cancel_dump_traceback_later()
        _flush_c_stdio()
        if use_dist:
            dist.barrier()                  # the other ranks have emptied theirs
        print(json.dumps(out), flush=True)
    elif use_dist:
        _flush_c_stdio()
        dist.barrier()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
