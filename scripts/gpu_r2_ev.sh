export TMPDIR=/tmp
mkdir -p gpurun_out
PL=$PWD/foundationpose_amd/csrc/libfp_amd_profile.so
FP_AMD_LIB=$PL python scripts/dbg_conv1.py 2>&1 | grep -v amdgpu > gpurun_out/r02_h_conv1_phase_timers.log; cat gpurun_out/r02_h_conv1_phase_timers.log
FP_AMD_LIB=$PL python scripts/dbg_conv_sw.py 2>&1 | grep -v amdgpu > gpurun_out/r02_h_conv_sw_phase_timers.log; cat gpurun_out/r02_h_conv_sw_phase_timers.log
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r02_h_pmc_conv1_sq -o k -- python scripts/bench_conv1.py > /dev/null 2>&1
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES --output-format csv -d gpurun_out/r02_h_pmc_conv1_grbm -o k -- python scripts/bench_conv1.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("sq", "grbm"):
    fs = glob.glob(f"gpurun_out/r02_h_pmc_conv1_{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no file"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "conv7x7" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: round(sum(v[-20:]) / len(v[-20:])) for k, v in acc.items()})
PY
