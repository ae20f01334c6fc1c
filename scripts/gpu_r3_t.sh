# round 3, call T (last GPU seconds of the round): the default bench command at HEAD (fused out_proj + LayerNorm on by default)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 45 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r3t_bench.json 2> /dev/null; python scripts/show_bench_kernels.py gpurun_out/r3t_bench.json
