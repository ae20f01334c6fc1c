export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
timeout 500 python bench.py > gpurun_out/bench2.json 2> gpurun_out/bench2.err; tail -5 gpurun_out/bench2.err; cut -c1-3000 gpurun_out/bench2.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench2_prof.log 2>&1
find gpurun_out/prof2 -type f | head
f=$(find gpurun_out/prof2 -name "*kernel_stats.csv" | head -1); head -25 "$f"
