# round 2, GPU call B: kernel tests, per-layer GEMM, bench + cpu baseline, rocprof kernel stats, PMC traffic / MFMA busy,
# config 5 (tracking sequence), multi-GPU launch behaviour on a 1-GPU box
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r02b
echo "== kernels"; timeout 400 python -m pytest tests/test_gpu_amp.py -q -k "not 252" > gpurun_out/${T}_kernels.log 2>&1; tail -12 gpurun_out/${T}_kernels.log
echo "== per-layer igemm"; timeout 200 python scripts/bench_igemm.py > gpurun_out/${T}_igemm.log 2>&1; tail -14 gpurun_out/${T}_igemm.log
echo "== bench"; timeout 500 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -2 gpurun_out/${T}_bench.err; cut -c1-300 gpurun_out/${T}_bench.json
echo "== rocprof kernel stats"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_prof.log 2>&1
head -14 gpurun_out/${T}_prof/bench_kernel_stats.csv | cut -c1-170
echo "== PMC traffic"; timeout 220 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}_pmc_fetch -o k -- python scripts/run_kernels.py > /dev/null 2>&1
timeout 220 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}_pmc_write -o k -- python scripts/run_kernels.py > /dev/null 2>&1
python scripts/pmc_traffic.py gpurun_out/${T}_pmc_fetch/k_counter_collection.csv gpurun_out/${T}_pmc_write/k_counter_collection.csv gpurun_out/${T}_traffic.json | tr -d '\n' | cut -c1-1500; echo
echo "== PMC MFMA busy"
for l in 512 256 128; do
FP_LAYER=$l timeout 122 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/${T}_pmc_sq_$l -o k -- python scripts/one_conv.py > /dev/null 2>&1
FP_LAYER=$l timeout 122 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES --output-format csv -d gpurun_out/${T}_pmc_grbm_$l -o k -- python scripts/one_conv.py > /dev/null 2>&1
python - $l $T <<'PY'
import csv,sys,collections,glob
l,T=sys.argv[1],sys.argv[2]
for d in (f"gpurun_out/{T}_pmc_sq_{l}", f"gpurun_out/{T}_pmc_grbm_{l}"):
    f=glob.glob(d+"/*counter_collection.csv")
    if not f: print(d,"none"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "conv_sw" in r["Kernel_Name"] or "igemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    print(l, {k:(v[-1][0], round(v[-1][1]/1e3,1)) for k,v in agg.items()})
PY
done
echo "== config 5"; timeout 400 python scripts/bench_track.py > gpurun_out/${T}_track.json 2> gpurun_out/${T}_track.err; tail -2 gpurun_out/${T}_track.err; cut -c1-900 gpurun_out/${T}_track.json
echo "== multi-GPU launch on a 1-GPU box"; python bench.py --gpus 2 > gpurun_out/${T}_gpus2.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/${T}_gpus2.log
FP_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_rccl1.json 2> gpurun_out/${T}_rccl1.err; echo "rccl rc=$?"; cut -c1-250 gpurun_out/${T}_rccl1.json
FP_BENCH_FORCE_DIST=1 timeout 300 python bench.py --mode hypothesis --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_hyp1.json 2> gpurun_out/${T}_hyp1.err; echo "hyp rc=$?"; cut -c1-250 gpurun_out/${T}_hyp1.json
