# round 3, call L: SQ counters of the raster / crop stage kernels (what bounds them), separate PMC passes over scripts/run_kernels.py
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
FP_REPS=4 timeout 220 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d gpurun_out/r3l_pmc_sq -o k -- python scripts/run_kernels.py > /dev/null 2>&1
FP_REPS=4 timeout 220 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r3l_pmc_grbm -o k -- python scripts/run_kernels.py > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob, json
out = {}
for d in ("gpurun_out/r3l_pmc_sq", "gpurun_out/r3l_pmc_grbm"):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        for key in ("k_raster", "k_warp", "k_vertex", "k_bin", "k_attention", "k_conv7x7"):
            if key in name:
                agg[key][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, c in agg.items():
        for cn, v in c.items():
            v = v[len(v) // 2:]                       # warm launches
            out.setdefault(k, {})[cn] = sum(x[0] for x in v) / len(v)
            out[k]["duration_us_under_profiler"] = sum(x[1] for x in v) / len(v) / 1e3
json.dump(out, open("gpurun_out/r3l_stage_counters.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(k, {a: round(b, 1) for a, b in v.items()})
PY
