# round 4, call A: graphs in the product (equality test + bench A/B), the prepared FFN tile (micro-bench + bench A/B + parity),
# the parity gates against the exactly-rounded yardstick, PMC passes over one serialized bench step (traffic + MFMA busy), smoke
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo "== $1 (t=$(( $(date +%s) - T0 )) s)"; }
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
el "graph equality"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -x -k "graphed_predict or sub_batches or graphed_tracker or shared_observed" > gpurun_out/r4a_pytest_graph.log 2>&1; tail -5 gpurun_out/r4a_pytest_graph.log | cut -c1-400
el "bench graph / eager / fused FFN"
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; python scripts/show_bench_kernels.py gpurun_out/r4a_bench.json
timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-graph > gpurun_out/r4a_bench_nograph.json 2> /dev/null; cut -c1-260 gpurun_out/r4a_bench_nograph.json
FP_AMD_FUSED_FFN=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4a_bench_ffn.json 2> gpurun_out/r4a_bench_ffn.err; python scripts/show_bench_kernels.py gpurun_out/r4a_bench_ffn.json
FP_AMD_FUSED_FFN=1 timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-graph > gpurun_out/r4a_bench_ffn_nograph.json 2> /dev/null; cut -c1-260 gpurun_out/r4a_bench_ffn_nograph.json
el "FFN micro-bench"
FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so timeout 120 python scripts/bench_linear_ln_mean.py > gpurun_out/r4a_ffn_micro.log 2>&1; cat gpurun_out/r4a_ffn_micro.log | cut -c1-300
FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so FP_LL_TILE=64 timeout 120 python scripts/bench_linear_ln.py > gpurun_out/r4a_ll_tile64.log 2>&1; tail -6 gpurun_out/r4a_ll_tile64.log | cut -c1-300
el "parity vs exact"
timeout 900 python -m pytest tests/test_gpu_amp.py -m gpu -q --timeout 600 --durations=5 -k "vs_exact" > gpurun_out/r4a_pytest_exact.log 2>&1; tail -30 gpurun_out/r4a_pytest_exact.log | cut -c1-600
el "parity vs exact, fused FFN"
FP_AMD_FUSED_FFN=1 FP_PARITY_REPORT=parity_amp_ffn.json timeout 600 python -m pytest tests/test_gpu_amp.py -m gpu -q --timeout 500 -k "refiner_252_teacher or contraction or plans_match" > gpurun_out/r4a_pytest_exact_ffn.log 2>&1; tail -30 gpurun_out/r4a_pytest_exact_ffn.log | cut -c1-600
el "PMC over one serialized step"
BENCH1="python bench.py --serialize --no-graph --steps 1 --warmup 1 --trace-markers --no-kernel-table --no-cpu-baseline"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r4a_pmc_fetch -o k -- $BENCH1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r4a_pmc_write -o k -- $BENCH1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r4a_pmc_mfma -o k -- $BENCH1 > /dev/null 2>&1
F=$(ls gpurun_out/r4a_pmc_fetch/*counter_collection.csv 2>/dev/null | head -1); W=$(ls gpurun_out/r4a_pmc_write/*counter_collection.csv 2>/dev/null | head -1); M=$(ls gpurun_out/r4a_pmc_mfma/*counter_collection.csv 2>/dev/null | head -1)
python scripts/pmc_step_traffic.py "$F" "$W" gpurun_out/r4a_traffic.json | tr -d '\n' | cut -c1-1500; echo
python scripts/pmc_step_mfma.py "$M" gpurun_out/r4a_mfma_busy.json
# keep the PMC csvs small: only the columns used
for f in "$F" "$W" "$M"; do python - "$f" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
keep = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
keep = [k for k in keep if rows and k in rows[0]]
w = csv.DictWriter(open(p, "w"), fieldnames=keep)
w.writeheader()
for r in rows:
    w.writerow({k: (r[k][:80] if k == "Kernel_Name" else r[k]) for k in keep})
PY
done
el "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-500
el "done"
