export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
for n in 126 252; do echo "N=$n"; FP_N=$n FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so timeout 120 python scripts/raster_phases.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4e_raster_phases.log; cat gpurun_out/r4e_raster_phases.log
FP_N=126 FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4e_prof -o rp -- python scripts/raster_phases.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4e_prof/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]:
    print(r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
