# round 3, call D: concurrent-mode kernel trace of the default bench + the re-split chain / scorer tests + new-tile policy tests
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d_prof -o bench -- python bench.py --steps 5 --warmup 3 --trace-markers --no-kernel-table --no-cpu-baseline > gpurun_out/r3d_bench_traced.json 2> gpurun_out/r3d_bench_traced.err
python scripts/concurrent_roofline.py gpurun_out/r3d_prof/bench_kernel_trace.csv gpurun_out/r3d_bench_traced.json > gpurun_out/r3d_concurrent_roofline.json; head -c 3000 gpurun_out/r3d_concurrent_roofline.json
rm -f gpurun_out/r3d_prof/bench_kernel_trace.csv.keep; python - <<'PY'
# keep the trace small: only the timed region's rows
import csv
rows=list(csv.DictReader(open('gpurun_out/r3d_prof/bench_kernel_trace.csv')))
marks=[r for r in rows if 'k_depth_to_xyz' in r['Kernel_Name'] and r['Grid_Size_X']=='64' and r['Grid_Size_Y']=='4']
t0,t1=int(marks[-2]['Start_Timestamp']),int(marks[-1]['End_Timestamp'])
keep=[r for r in rows if t0<=int(r['Start_Timestamp'])<=t1]
w=csv.DictWriter(open('gpurun_out/r3d_timed_region_kernel_trace.csv','w'),fieldnames=['Stream_Id','Kernel_Name','Start_Timestamp','End_Timestamp','Grid_Size_X','Workgroup_Size_X','VGPR_Count','LDS_Block_Size'])
w.writeheader()
for r in keep: w.writerow({k:(r[k][:60] if k=='Kernel_Name' else r[k]) for k in w.fieldnames})
print(len(keep),'dispatches in the timed region')
PY
echo "trace seconds: $(( $(date +%s) - T0 ))"
timeout 200 python bench.py --steps 10 --warmup 3 > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; cut -c1-300 gpurun_out/r3d_bench.json
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_amp.py -m gpu -q --timeout 420 --durations=8 \
  -k "free_running or scorer_252 or rccl or policy or eight" > gpurun_out/r3d_pytest.log 2>&1; tail -25 gpurun_out/r3d_pytest.log | cut -c1-300
echo "total seconds: $(( $(date +%s) - T0 ))"
