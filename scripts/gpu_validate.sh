export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu22.log 2>&1; tail -3 gpurun_out/pytest_gpu22.log
timeout 220 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py > gpurun_out/bench22.json 2> gpurun_out/bench22.err; tail -2 gpurun_out/bench22.err; cut -c1-400 gpurun_out/bench22.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof22 -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench22_prof.log 2>&1
head -8 gpurun_out/prof22/bench_kernel_stats.csv | cut -c1-160
timeout 220 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc22_fetch -o k -- python scripts/run_kernels.py > /dev/null 2>&1
timeout 220 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc22_write -o k -- python scripts/run_kernels.py > /dev/null 2>&1
python scripts/pmc_traffic.py gpurun_out/pmc22_fetch/k_counter_collection.csv gpurun_out/pmc22_write/k_counter_collection.csv gpurun_out/traffic22.json > /dev/null
for l in 512 256; do
FP_LAYER=$l timeout 122 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc22_sq_$l -o k -- python scripts/one_conv.py > /dev/null 2>&1
FP_LAYER=$l timeout 122 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES --output-format csv -d gpurun_out/pmc22_grbm_$l -o k -- python scripts/one_conv.py > /dev/null 2>&1
python - $l <<'PY'
import csv,sys,collections,glob
l=sys.argv[1]
for d in ("gpurun_out/pmc22_sq_"+l, "gpurun_out/pmc22_grbm_"+l):
    f=glob.glob(d+"/*counter_collection.csv")
    if not f: print(d,"none"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "igemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    print(l, {k:(v[-1][0], round(v[-1][1]/1e3,1)) for k,v in agg.items()})
PY
done
