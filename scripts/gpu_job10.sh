export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu10.log 2>&1; tail -4 gpurun_out/pytest_gpu10.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 500 python bench.py > gpurun_out/bench10.json 2> gpurun_out/bench10.err; tail -3 gpurun_out/bench10.err; cut -c1-1200 gpurun_out/bench10.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof10 -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench10_prof.log 2>&1
head -12 gpurun_out/prof10/bench_kernel_stats.csv | cut -c1-200
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc10_fetch -o k -- python scripts/run_kernels.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc10_write -o k -- python scripts/run_kernels.py > /dev/null 2>&1
python scripts/pmc_traffic.py gpurun_out/pmc10_fetch/k_counter_collection.csv gpurun_out/pmc10_write/k_counter_collection.csv gpurun_out/traffic.json | head -60
