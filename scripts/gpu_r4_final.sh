# round 4, call H (512 x 128 conv tiles on every layer): the whole GPU suite at HEAD + the bench records (default command, --serialize under rocprofv3 --stats, PMC passes)
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo "== $1 (t=$(( $(date +%s) - T0 )) s)"; }
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
el "pytest -m gpu"
timeout 1500 python -m pytest tests/ -m gpu -q --timeout 900 --durations=8 > gpurun_out/r4h_pytest_gpu.log 2>&1; tail -16 gpurun_out/r4h_pytest_gpu.log | cut -c1-300
el "bench"
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err; python scripts/show_bench_kernels.py gpurun_out/r4h_bench.json
el "bench --serialize under rocprofv3 --stats"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4h_prof -o bench -- python bench.py --serialize --no-graph --steps 5 --warmup 2 --no-kernel-table --no-cpu-baseline > gpurun_out/r4h_bench_serialize.json 2> /dev/null
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4h_prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
w = csv.DictWriter(open("gpurun_out/r4h_bench_serialize_kernel_stats.csv", "w"), fieldnames=list(rows[0].keys()))
w.writeheader()
for r in rows[:40]:
    r["Name"] = r["Name"][:90]
    w.writerow(r)
for r in rows[:10]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf gpurun_out/r4h_prof
el "PMC over one serialized step"
BENCH1="python bench.py --serialize --no-graph --steps 1 --warmup 1 --trace-markers --no-kernel-table --no-cpu-baseline"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r4h_pmc_fetch -o k -- $BENCH1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r4h_pmc_write -o k -- $BENCH1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r4h_pmc_mfma -o k -- $BENCH1 > /dev/null 2>&1
F=$(ls gpurun_out/r4h_pmc_fetch/*counter_collection.csv | head -1); W=$(ls gpurun_out/r4h_pmc_write/*counter_collection.csv | head -1); M=$(ls gpurun_out/r4h_pmc_mfma/*counter_collection.csv | head -1)
python scripts/pmc_step_traffic.py "$F" "$W" gpurun_out/r4h_traffic.json > /dev/null; python scripts/pmc_step_mfma.py "$M" gpurun_out/r4h_mfma_busy.json | head -8
for f in "$F" "$W" "$M"; do python - "$f" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
keep = [k for k in ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"] if rows and k in rows[0]]
w = csv.DictWriter(open(p, "w"), fieldnames=keep)
w.writeheader()
for r in rows:
    w.writerow({k: (r[k][:80] if k == "Kernel_Name" else r[k]) for k in keep})
PY
done
el "config 5 + smoke"
timeout 300 python scripts/bench_track.py > gpurun_out/r4h_track_config5.json 2> /dev/null; cut -c1-400 gpurun_out/r4h_track_config5.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-400
el "done"
