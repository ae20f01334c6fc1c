# tests + smoke + bench + kernel stats (no PMC passes): the closing check of a session
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.log 2>&1; tail -2 gpurun_out/pytest_gpu_final.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-330 gpurun_out/bench_final.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_final_prof.log 2>&1
head -8 gpurun_out/prof_final/bench_kernel_stats.csv | cut -c1-140
