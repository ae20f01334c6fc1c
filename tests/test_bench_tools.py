"""CPU tests of the measurement helpers: interval union of scripts/concurrent_roofline.py (the busy time of a kernel family in
the two-stream timed mode), its reduction of a synthetic kernel trace, and bench.ClockSampler without an amdgpu hwmon."""
import csv
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_interval_union():
    import concurrent_roofline as cr
    assert cr.union([]) == 0
    assert cr.union([(0, 10)]) == 10
    assert cr.union([(0, 10), (5, 15)]) == 15            # overlapping launches count once
    assert cr.union([(0, 10), (10, 12), (20, 30)]) == 22  # touching intervals merge, gaps do not
    assert cr.union([(5, 6), (0, 10), (2, 3)]) == 10      # nested, unsorted


def test_concurrent_roofline_on_a_synthetic_trace(tmp_path, capsys):
    import concurrent_roofline as cr
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y"]
    rows = [("k_depth_to_xyz(float const*)", 0, 10, 64, 4),                       # marker in
            ("void (anonymous namespace)::k_conv_sw<256, 256, 4>(IgemmParams)", 1000, 2000, 1, 1),
            ("void k_igemm_pp<256, 256, 4, 0>(IgemmParams)", 1500, 2500, 1, 1),   # overlaps the first GEMM on another stream
            ("k_raster(fp_mesh)", 2500, 3000, 1, 1),
            ("k_depth_to_xyz(float const*)", 9990, 10000, 640, 480),             # a real call: not a marker
            ("k_depth_to_xyz(float const*)", 4000, 4010, 64, 4)]                 # marker out
    trace = tmp_path / "trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        w.writerows(rows)
    bench = tmp_path / "bench.json"
    bench.write_text(json.dumps(dict(steps=1, ms_per_step=1.0, config=dict(hypotheses_per_gpu=252, refine_iterations=5), clock=None)))
    cr.main(str(trace), str(bench))
    out = json.loads(capsys.readouterr().out)
    g = out["families"]["gemm"]
    assert g["launches"] == 2 and abs(g["sum_ms"] - 2000e-6) < 1e-12 and abs(g["busy_ms"] - 1500e-6) < 1e-12
    assert out["families"]["raster"]["launches"] == 1 and out["families"]["other"]["launches"] == 0
    assert abs(out["region_wall_ms"] - 3990e-6) < 1e-12
    rc = out["roofline_concurrent"]
    assert abs(rc["mean_concurrency"] - 2000 / 1500) < 1e-9 and rc["bound"] == "mfma"


def test_clock_sampler_without_hwmon():
    sys.path.insert(0, ROOT)
    import bench
    c = bench.ClockSampler(0)
    with c:
        pass
    s = c.summary()
    assert "sclk_MHz_mean" in s and (s["samples"] == 0 or s["sclk_MHz_mean"] > 0)


def test_packed_fp32_build_check_catches_a_packed_kernel(tmp_path):
    """csrc/check_no_pk_f32.py (run by `make`): passes on the product library, and FAILS on a library whose device code
    contains v_pk_*_f32 -- a two-float add that hipcc packs at -O3 without -fno-slp-vectorize"""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or shutil.which("python3") is None:
        import pytest
        pytest.skip("needs hipcc")
    chk = os.path.join(ROOT, "foundationpose_amd", "csrc", "check_no_pk_f32.py")
    lib = os.path.join(ROOT, "foundationpose_amd", "csrc", "libfp_amd.so")
    assert subprocess.run([sys.executable, chk, lib], capture_output=True).returncode == 0
    src = tmp_path / "pk.hip"
    src.write_text('#include <hip/hip_runtime.h>\n'
                   '__global__ void k(const float2* a, const float2* b, float2* c) { const int i = threadIdx.x; float2 x = a[i], y = b[i];\n'
                   '  c[i] = make_float2(x.x * y.x + x.x, x.y * y.y + x.y); }\n'
                   'extern "C" void launch(const float2* a, const float2* b, float2* c) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c); }\n')
    so = tmp_path / "libpk.so"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", str(src), "-o", str(so)])
    r = subprocess.run([sys.executable, chk, str(so)], capture_output=True, text=True)
    assert r.returncode != 0 and "packed-fp32" in (r.stderr + r.stdout), (r.stdout, r.stderr)
    # round 5: a library the check cannot look into (here: no device code at all) FAILS instead of warning; the skip is opt-in
    host = tmp_path / "host.c"
    host.write_text("int f(void) { return 1; }\n")
    hso = tmp_path / "libhost.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(host), "-o", str(hso)])
    r = subprocess.run([sys.executable, chk, str(hso)], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "FP_AMD_ALLOW_UNCHECKED"})
    assert r.returncode != 0 and "could not run" in (r.stderr + r.stdout), (r.stdout, r.stderr)
    r = subprocess.run([sys.executable, chk, str(hso)], capture_output=True, text=True, env=dict(os.environ, FP_AMD_ALLOW_UNCHECKED="1"))
    assert r.returncode == 0 and "skipped" in r.stderr


def test_bench_respawn_builds_the_launcher_command_line(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run, one rank per GPU on 127.0.0.1
    (the driver's own launch line for N > 1); no 8-GPU node was ever available to the builder, so the command line is pinned here"""
    import argparse
    import subprocess
    import torch
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--mode", "hypothesis"])
    try:
        bench._respawn(argparse.Namespace(gpus=8))
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "3", "--mode", "hypothesis"]      # every flag reaches every rank
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                                 # dmabuf IPC: RCCL needs it on this host driver


def test_bench_refuses_more_ranks_than_gpus():
    """on a node with fewer GPUs than --gpus the bench exits non-zero with the documented message instead of running fewer ranks
    (this container has no GPU at all; on the 1-GPU box: profiles/r02_g_bench_gpus2_on_1gpu_box.log)"""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this node has 8 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    assert "--gpus 8 requested but this node exposes" in r.stderr and "refusing to run fewer ranks" in r.stderr
    # under a launcher the world size has to match
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in r.stderr
