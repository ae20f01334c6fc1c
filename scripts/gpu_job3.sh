export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder or warp or fp16 or golden" > gpurun_out/pytest_gpu3.log 2>&1; tail -25 gpurun_out/pytest_gpu3.log
timeout 300 python scripts/bench_igemm.py > gpurun_out/bench_igemm.log 2>&1; cat gpurun_out/bench_igemm.log | tail -40
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -3 gpurun_out/bench3.err; cut -c1-2500 gpurun_out/bench3.json
