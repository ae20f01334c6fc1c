export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -o bench -- python bench.py > gpurun_out/bench1.log 2>&1
grep '^{' gpurun_out/bench1.log | cut -c1-2500
find gpurun_out/prof1 -type f | head
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); head -30 "$f"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o k -- python scripts/run_kernels.py > gpurun_out/pmc_fetch.log 2>&1; tail -2 gpurun_out/pmc_fetch.log
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o k -- python scripts/run_kernels.py > gpurun_out/pmc_write.log 2>&1; tail -2 gpurun_out/pmc_write.log
find gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head; f=$(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1); head -5 "$f"
