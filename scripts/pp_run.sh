export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "igemm or hip_encoder or fp16_plans or graph" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline 2> gpurun_out/bench_x.err | cut -c1-600
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_x_prof.log 2>&1
head -14 gpurun_out/prof_x/bench_kernel_stats.csv | cut -c1-150
