# the round's closing records: TAG=r06_final bash scripts/gpu_r06_final.sh   (gpu.sh tasks + the concurrent-roofline trace)
export TMPDIR=/tmp
TAG=${TAG:-r06_final}
O=gpurun_out; mkdir -p $O
TAG=$TAG bash scripts/gpu.sh sane tests smoke bench prof pmc 2>&1 | tee $O/${TAG}_gpu_sh.log | cut -c1-400
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace -o bench -- python bench.py --trace-markers --no-kernel-table --no-cpu-baseline --no-extras > $O/${TAG}_bench_traced.json 2> /dev/null
T=$(ls $O/${TAG}_trace/*kernel_trace.csv $O/${TAG}_trace/*/*kernel_trace.csv 2>/dev/null | head -1)
python scripts/concurrent_roofline.py "$T" $O/${TAG}_bench_traced.json > $O/${TAG}_concurrent_roofline.json 2> /dev/null; cut -c1-600 $O/${TAG}_concurrent_roofline.json
rm -rf $O/${TAG}_trace
