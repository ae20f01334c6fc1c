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
